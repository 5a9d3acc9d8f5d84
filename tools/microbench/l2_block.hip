// Microbenchmark: can the two LAST passes of the wideband FFT share their intermediate array through an XCD's L2?
//
// After the first (strided) pass, FFT_N (N = 600 x 400 000) decomposes into 600 independent transforms of contiguous
// 3.2 MB blocks, each of which is two passes (column tiles, then row tiles).  If ONE XCD runs both passes of a block and
// keeps the intermediate in a 3.2 MB scratch that it rewrites for every block, the scratch lines stay dirty in that XCD's
// 4 MiB L2 and never need to reach HBM: 16 bytes per point instead of 32 for the two passes.  This file measures
// whether the memory system plays along, with tile copies in the FFT's shapes (no arithmetic):
//
//   two launches   P1: in -> mid (column tiles: 16 lines x ROWS rows), P2: mid -> out (row tiles: 16 rows x COLS)
//   fused          one persistent launch; every workgroup reads its XCD from HW_REG_XCC_ID and joins that XCD's team;
//                  a team takes tickets from its own counter: P1 tiles of block b (in -> scratch[xcd]), then P2 tiles of
//                  block b (scratch[xcd] -> out).  P1 loads are issued BEFORE the wait for "scratch free", P2 stores are
//                  not waited for, so only the L2 traffic is serialised.  Counters are touched by one XCD only, with
//                  atomics that execute in that XCD's L2; payload loads from the scratch bypass L1 (sc1).
//
// Every spin is bounded (a flag is raised instead of hanging the box).  Output: GB/s of in + out bytes, a full check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
struct Ctl { unsigned ticket, err, members, pad0[29]; unsigned done1[8], pad1[24]; unsigned done2[8], pad2[24]; unsigned pad3[32]; };   // 512 bytes per XCD, counters on their own lines

__device__ inline unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu; }

__device__ inline unsigned poll_l2(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// POLL 0: sc1 load; 1: returning atomic OR 0 (executes in the L2); 2: sc0 sc1 load
template <int POLL> __device__ inline unsigned poll(unsigned* p) {
    if (POLL == 0) return poll_l2(p);
    if (POLL == 1) { unsigned v; asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(0u) : "memory"); return v; }
    unsigned v; asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v;
}

// One lane of the calling wave issues the atomic; exec is narrowed inside the asm so that the compiler sees no divergent
// branch (a divergent `if (tid == 0)` at the top of the ticket loop is jump-threaded into two loops and the barriers
// of lane 0 and of lanes 1..63 of wave 0 no longer pair up: the first version of this file hung exactly there).
__device__ inline unsigned wave_fetch_add1(unsigned* p) {
    unsigned v = 0;
    asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)\n\ts_mov_b64 exec, s[20:21]"
                 : "+v"(v) : "v"(p), "v"(1u) : "memory", "s20", "s21");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ inline void wave_add1(unsigned* p) {
    asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %1, off\n\ts_mov_b64 exec, s[20:21]"
                 :: "v"(p), "v"(1u) : "memory", "s20", "s21");
}
__device__ inline void wave_or(unsigned* p, unsigned bits) {
    asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_or %0, %1, off\n\ts_mov_b64 exec, s[20:21]"
                 :: "v"(p), "v"(bits) : "memory", "s20", "s21");
}

template <int ROWS, int COLS, int T> struct Shape {
    static constexpr int kP1Tiles = COLS / 16;
    static constexpr int kP2Tiles = (ROWS + 15) / 16;
    static constexpr int kBlkF4 = ROWS * COLS / 2;
    static constexpr int kK1 = (ROWS * 8 + T - 1) / T;         // f4 per thread, column tile
    static constexpr int kK2 = (16 * COLS / 2 + T - 1) / T;    // f4 per thread, row tile
};

// element index (f4) of a column tile's e-th f4: row e / 8, 8 f4 (= 16 complex) per row segment
template <int COLS> __device__ inline int p1_index(int tile, int e) { return (e >> 3) * (COLS / 2) + tile * 8 + (e & 7); }

template <int ROWS, int COLS, int T, bool P2>
__global__ __launch_bounds__(T) void k_pass(const f4* __restrict__ src, f4* __restrict__ dst) {
    using S = Shape<ROWS, COLS, T>;
    extern __shared__ char lds[];
    constexpr int tiles = P2 ? S::kP2Tiles : S::kP1Tiles;
    const unsigned gx = gridDim.x, x = blockIdx.x;
    const unsigned tix = (gx & 7u) ? x : (x & 7u) * (gx >> 3) + (x >> 3);
    const size_t blk = tix / tiles; const int tile = tix % tiles;
    const f4* s = src + blk * S::kBlkF4; f4* d = dst + blk * S::kBlkF4;
    if (!P2) {
        f4 v[S::kK1];
#pragma unroll
        for (int k = 0; k < S::kK1; ++k) { int e = threadIdx.x + T * k; e = e < ROWS * 8 ? e : 0; v[k] = s[p1_index<COLS>(tile, e)]; }
#pragma unroll
        for (int k = 0; k < S::kK1; ++k) { int e = threadIdx.x + T * k; if (e < ROWS * 8) __builtin_nontemporal_store(v[k], &d[p1_index<COLS>(tile, e)]); }
    } else {
        const int base = tile * 16 * COLS / 2, lim = (ROWS - tile * 16 < 16 ? ROWS - tile * 16 : 16) * COLS / 2;
        f4 v[S::kK2];
#pragma unroll
        for (int k = 0; k < S::kK2; ++k) { int e = threadIdx.x + T * k; e = e < lim ? e : 0; v[k] = s[base + e]; }
#pragma unroll
        for (int k = 0; k < S::kK2; ++k) { int e = threadIdx.x + T * k; if (e < lim) __builtin_nontemporal_store(v[k], &d[base + e]); }
    }
    if (lds[0] == 77 && threadIdx.x == 12345) dst[0].x = 0.f;   // keeps the allocation
}

// LOADMODE 0: sc1 loads from the scratch (bypass L1); 1: agent acquire (buffer_inv sc1) + plain loads
template <int ROWS, int COLS, int T, int LOADMODE, bool NT_IN, int POLL>
__global__ __launch_bounds__(T) void k_fused(const f4* __restrict__ in, f4* __restrict__ out, f4* scratch,
                                             Ctl* ctl, int blocks_per_xcd, unsigned spin_max, int slots) {
    using S = Shape<ROWS, COLS, T>;
    extern __shared__ char lds[];
    __shared__ unsigned s_t;
    const unsigned x = xcc_id() & 7u;
    Ctl* c = ctl + x;
    f4* scr0 = scratch + (size_t)x * slots * S::kBlkF4;
    const int tid = threadIdx.x;
    constexpr unsigned per_block = S::kP1Tiles + S::kP2Tiles;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave == 0) wave_add1(&c->members);
    for (;;) {
        if (wave == 0) s_t = wave_fetch_add1(&c->ticket);
        __syncthreads();
        const unsigned t = __builtin_amdgcn_readfirstlane(s_t);
        __syncthreads();
        // ticket order: P1(0), then [P1(g), P2(g-1)] for g = 1 .. nblk (P1(nblk) does not exist): a waiting P2 always has
        // a whole phase of other tiles in front of it; P1(b) waits for P2(b - slots), which precedes it when slots >= 2
        unsigned bl, r; bool p1;
        if (slots == 1) { bl = t / per_block; r = t % per_block; p1 = r < (unsigned)S::kP1Tiles; if (!p1) r -= S::kP1Tiles; if (bl >= (unsigned)blocks_per_xcd) break; }
        else if (t < (unsigned)S::kP1Tiles) { bl = 0; r = t; p1 = true; }
        else {
            const unsigned u = t - S::kP1Tiles, g = u / per_block + 1; r = u % per_block;
            if (g > (unsigned)blocks_per_xcd) break;
            p1 = r < (unsigned)S::kP1Tiles;
            if (p1) { bl = g; if (g == (unsigned)blocks_per_xcd) continue; } else { bl = g - 1; r -= S::kP1Tiles; }
        }
        const size_t blk = x + 8 * (size_t)bl;
        const unsigned slot = bl % slots, use = bl / slots;
        f4* scr = scr0 + (size_t)slot * S::kBlkF4;
        if (p1) {
            const int tile = r;
            const f4* s = in + blk * S::kBlkF4;
            f4 v[S::kK1];
#pragma unroll
            for (int k = 0; k < S::kK1; ++k) {
                int e = tid + T * k; e = e < ROWS * 8 ? e : 0;
                v[k] = NT_IN ? __builtin_nontemporal_load(&s[p1_index<COLS>(tile, e)]) : s[p1_index<COLS>(tile, e)];
            }
            if (wave == 0) {     // scratch free: every row tile of the previous block has been read
                unsigned n = 0;
                while (__builtin_amdgcn_readfirstlane(poll<POLL>(&c->done2[slot])) < use * S::kP2Tiles) { __builtin_amdgcn_s_sleep(2); if (++n > spin_max || __builtin_amdgcn_readfirstlane(poll<POLL>(&c->err))) { wave_or(&c->err, 1u); break; } }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < S::kK1; ++k) { int e = tid + T * k; if (e < ROWS * 8) scr[p1_index<COLS>(tile, e)] = v[k]; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (wave == 0) wave_add1(&c->done1[slot]);
        } else {
            const int tile = r;
            const int base = tile * 16 * COLS / 2, lim = (ROWS - tile * 16 < 16 ? ROWS - tile * 16 : 16) * COLS / 2;
            if (wave == 0) {
                unsigned n = 0;
                while (__builtin_amdgcn_readfirstlane(poll<POLL>(&c->done1[slot])) < (use + 1) * S::kP1Tiles) { __builtin_amdgcn_s_sleep(2); if (++n > spin_max || __builtin_amdgcn_readfirstlane(poll<POLL>(&c->err))) { wave_or(&c->err, 2u); break; } }
            }
            __syncthreads();
            f4 v[S::kK2];
            if (LOADMODE == 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
                for (int k = 0; k < S::kK2; ++k) { int e = tid + T * k; e = e < lim ? e : 0; v[k] = scr[base + e]; }
            } else {
#pragma unroll
                for (int k = 0; k < S::kK2; ++k) {
                    int e = tid + T * k; e = e < lim ? e : 0;
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[k]) : "v"(&scr[base + e]) : "memory");
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (wave == 0) wave_add1(&c->done2[slot]);
            f4* d = out + blk * S::kBlkF4;
#pragma unroll
            for (int k = 0; k < S::kK2; ++k) { int e = tid + T * k; if (e < lim) __builtin_nontemporal_store(v[k], &d[base + e]); }
        }
    }
    if (lds[0] == 77 && tid == 12345) out[0].x = 0.f;
}

__global__ void k_fill(f4* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned a = (unsigned)(i * 2654435761u), b = (unsigned)(i >> 7);
        p[i] = f4{__uint_as_float(0x3f800000u | (a & 0x7fffffu)), __uint_as_float(0x3f800000u | (b & 0x7fffffu)), (float)(i & 1023), 1.f};
    }
}
__global__ void k_cmp(const f4* a, const f4* b, size_t n, unsigned long long* bad) {
    unsigned long long m = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const f4 u = a[i], v = b[i];
        m += !(u.x == v.x && u.y == v.y && u.z == v.z && u.w == v.w);
    }
    if (m) atomicAdd(bad, m);
}

template <int ROWS, int COLS, int T, int LOADMODE, bool NT_IN, int POLL = 0>
void run(const char* name, int nblocks, int wg_per_cu, size_t lds_bytes, int reps, int slots = 1) {
    using S = Shape<ROWS, COLS, T>;
    const size_t nf4 = (size_t)nblocks * S::kBlkF4, bytes = nf4 * 16;
    f4 *in, *mid, *out, *scratch; Ctl* ctl; unsigned long long* bad;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&mid, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMalloc(&scratch, (size_t)8 * slots * S::kBlkF4 * 16)); CK(hipMalloc(&ctl, 8 * sizeof(Ctl))); CK(hipMalloc(&bad, 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, in, nf4);
    CK(hipMemset(out, 0, bytes)); CK(hipMemset(mid, 0, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    CK(hipFuncSetAttribute((const void*)k_pass<ROWS, COLS, T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    CK(hipFuncSetAttribute((const void*)k_pass<ROWS, COLS, T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    CK(hipFuncSetAttribute((const void*)k_fused<ROWS, COLS, T, LOADMODE, NT_IN, POLL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    printf("[%s] filled\n", name); CK(hipDeviceSynchronize()); printf("[%s] synced\n", name);
    // two launches
    double two = 0;
    for (int r = -1; r < reps; ++r) {
        if (r == 0) CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_pass<ROWS, COLS, T, false>), dim3(nblocks * S::kP1Tiles), dim3(T), lds_bytes, 0, in, mid);
        hipLaunchKernelGGL((k_pass<ROWS, COLS, T, true>), dim3(nblocks * S::kP2Tiles), dim3(T), lds_bytes, 0, mid, out);
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    two = ms / reps; printf("[%s] two launches done %.3f ms\n", name, two);
    CK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(k_cmp, dim3(4096), dim3(256), 0, 0, in, out, nf4, bad);
    unsigned long long h_bad2 = 0; CK(hipMemcpy(&h_bad2, bad, 8, hipMemcpyDeviceToHost));
    // fused
    CK(hipMemset(out, 0, bytes));
    const int grid = 256 * wg_per_cu;
    double fused = 0; Ctl h[8]; unsigned err = 0;
    for (int r = -1; r < reps; ++r) {
        CK(hipMemsetAsync(ctl, 0, 8 * sizeof(Ctl), 0));
        if (r == 0) CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_fused<ROWS, COLS, T, LOADMODE, NT_IN, POLL>), dim3(grid), dim3(T), lds_bytes, 0, in, out, scratch, ctl,
                           nblocks / 8, 300000u, slots);
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    fused = ms / reps;
    CK(hipMemcpy(h, ctl, sizeof(h), hipMemcpyDeviceToHost));
    CK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(k_cmp, dim3(4096), dim3(256), 0, 0, in, out, nf4, bad);
    unsigned long long h_bad = 0; CK(hipMemcpy(&h_bad, bad, 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) err |= h[i].err;
    printf("%-28s block %4d x %4d (%.2f MB) x %d, %d wg/cu lds %zu slots %d: two launches %.3f ms = %6.0f GB/s of in+out (bad %llu) | fused %.3f ms = %6.0f GB/s (bad %llu, err %u) members",
           name, ROWS, COLS, S::kBlkF4 * 16 / 1e6, nblocks, wg_per_cu, lds_bytes, slots, two, 2.0 * bytes / two / 1e6, h_bad2, fused,
           2.0 * bytes / fused / 1e6, h_bad, err);
    for (int i = 0; i < 8; ++i) printf(" %u(t%u a%u b%u e%u)", h[i].members, h[i].ticket, h[i].done1[0], h[i].done2[0], h[i].err);
    printf("\n"); fflush(stdout);
    CK(hipFree(in)); CK(hipFree(mid)); CK(hipFree(out)); CK(hipFree(scratch)); CK(hipFree(ctl)); CK(hipFree(bad));
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 3;
    setvbuf(stdout, nullptr, _IONBF, 0);
    run<320, 320, 1024, 0, true, 0>("tiny 1 slot", 16, 2, 81920, 1, 1);
    run<320, 320, 1024, 0, true, 0>("tiny 2 slots", 16, 2, 81920, 1, 2);
    run<320, 320, 1024, 0, true, 0>("tiny 3 slots", 24, 2, 81920, 1, 3);
    if (argc > 2 && !strcmp(argv[2], "prof")) {
        run<625, 640, 1024, 0, true>("3.2MB", 600, 2, 81920, reps, 1);
        run<625, 640, 1024, 0, true>("3.2MB", 600, 2, 81920, reps, 3);
        run<200, 256, 1024, 0, true>("0.4MB", 4800, 2, 81920, reps, 4);
        return 0;
    }
    if (argc > 2) return 0;
    // the wideband FFT's blocks: 625 x 640 complex = 3.2 MB, 600 of them, two 1024-thread workgroups per CU (80 KiB LDS each)
    for (int slots = 1; slots <= 4; ++slots) run<625, 640, 1024, 0, true>("3.2MB", 600, 2, 81920, reps, slots);
    for (int slots = 1; slots <= 4; ++slots) run<400, 512, 1024, 0, true>("1.6MB", 1200, 2, 81920, reps, slots);
    for (int slots = 1; slots <= 8; slots *= 2) run<320, 320, 1024, 0, true>("0.8MB", 2400, 2, 81920, reps, slots);
    for (int slots = 2; slots <= 8; slots *= 2) run<200, 256, 1024, 0, true>("0.4MB", 4800, 2, 81920, reps, slots);
    return 0;
}
