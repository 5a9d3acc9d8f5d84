// Internal helpers shared by the librcfm translation units (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <exception>
#include <string>

#include "rcfm.h"
#include "rcfm_tools.h"

namespace rcfm {

struct Error {
    int code;
    std::string msg;
};

void set_last_error(const std::string& msg);

#define RC_HIP(expr)                                                                     \
    do {                                                                                 \
        hipError_t rc_e_ = (expr);                                                       \
        if (rc_e_ != hipSuccess)                                                         \
            throw ::rcfm::Error{RCFM_ERR_RUNTIME,                                        \
                                std::string(#expr) + ": " + hipGetErrorString(rc_e_)};   \
    } while (0)

#define RC_REQUIRE(cond, code, text)                      \
    do {                                                  \
        if (!(cond)) throw ::rcfm::Error{(code), (text)}; \
    } while (0)

// Every extern "C" entry point runs its body through this: exceptions never
// cross the ABI, the message is kept for rcfm_last_error().
template <class F>
int guarded(F&& body) {
    try {
        body();
        return RCFM_OK;
    } catch (const Error& e) {
        set_last_error(e.msg);
        return e.code;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return RCFM_ERR_RUNTIME;
    } catch (...) {
        set_last_error("unknown failure");
        return RCFM_ERR_RUNTIME;
    }
}

// ---- placement control (rcfm_arena_*, include/rcfm.h) ----------------------------------------------------------
// Where hipMalloc puts a workspace moves the kernels that stream through it by several per cent
// (profiles/r04_k_placement.md).  A caller who wants ONE draw for a whole handle set creates an arena -- a few large
// device blocks, bump-allocated -- and binds it while the handles are created; every workspace of those handles of
// kArenaMinBytes or more then comes from the arena, for the handles' whole life (lazy growth included).  Arena use is
// opt-in per handle: only a tuner / demodulator handle adopts the bound arena, and a DeviceBuffer draws from an arena
// only while such a handle's entry point runs (ArenaScope) -- function-static scratch, plan caches and the other handle
// kinds always come from hipMalloc.
using Arena = ::rcfm_arena_s;
constexpr size_t kArenaMinBytes = (size_t)1 << 20;
Arena* current_arena();                         // thread-local: the arena of the handle whose entry point is running (ArenaScope), or null
Arena* arena_enter_handle();                    // a tuner / demodulator handle is being created: the arena the host bound (counted), or null
void arena_leave_handle(Arena* a);              // ... and destroyed
void* arena_take(Arena* a, size_t bytes);       // a piece of the arena (never null: the arena grows by whole blocks)
void arena_drop(Arena* a);                      // a piece is no longer used (the bytes return when the arena goes)
struct ArenaScope {                             // entry points of a handle run inside its arena
    Arena* prev;
    explicit ArenaScope(Arena* a);
    ~ArenaScope();
    ArenaScope(const ArenaScope&) = delete;
    ArenaScope& operator=(const ArenaScope&) = delete;
};

// Owning device allocation (handles own their workspaces; freed on destroy).
class DeviceBuffer {
   public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t bytes) { reset(bytes); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    DeviceBuffer(DeviceBuffer&& o) noexcept : p_(o.p_), bytes_(o.bytes_), arena_(o.arena_) {
        o.p_ = nullptr;
        o.bytes_ = 0;
        o.arena_ = nullptr;
    }
    DeviceBuffer& operator=(DeviceBuffer&& o) noexcept {
        if (this != &o) {
            release();
            p_ = o.p_;
            bytes_ = o.bytes_;
            arena_ = o.arena_;
            o.p_ = nullptr;
            o.bytes_ = 0;
            o.arena_ = nullptr;
        }
        return *this;
    }
    ~DeviceBuffer() { release(); }
    void reset(size_t bytes) {
        release();
        if (bytes) {
            Arena* a = bytes >= kArenaMinBytes ? current_arena() : nullptr;
            if (a) {
                p_ = arena_take(a, bytes);
                arena_ = a;
            } else {
                RC_HIP(hipMalloc(&p_, bytes));
            }
            bytes_ = bytes;
        }
    }
    // grow-only
    void reserve(size_t bytes) {
        if (bytes > bytes_) reset(bytes);
    }
    void upload(const void* host, size_t bytes) {
        reserve(bytes);
        if (bytes) RC_HIP(hipMemcpy(p_, host, bytes, hipMemcpyHostToDevice));
    }
    template <class T>
    T* as() const {
        return static_cast<T*>(p_);
    }
    void* get() const { return p_; }
    size_t bytes() const { return bytes_; }

   private:
    void release() {
        if (p_ && arena_) arena_drop(arena_);
        else if (p_) (void)hipFree(p_);
        p_ = nullptr;
        bytes_ = 0;
        arena_ = nullptr;
    }
    void* p_ = nullptr;
    size_t bytes_ = 0;
    Arena* arena_ = nullptr;
};

inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

}  // namespace rcfm
