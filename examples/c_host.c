/* A host without Python: librcfm.so driven through include/rcfm.h only.
 *
 * The reference's server loop (examples/multi_fm_server.py:86-106: take a one-second buffer, Tuner.load, per channel
 * Tuner.run -> WBFM.run, publish) in C: page-locked ring of input buffers -> rcfm_feeder (H2D copy of buffer i+1 under
 * the kernels of buffer i) -> rcfm_tuner_load -> rcfm_pipeline_run -> rcfm_gather_audio on a one-rank communicator ->
 * host.  Writes the input buffers and the audio to files so that tests/test_c_host.py can check them against the
 * oracle; then the same buffers through the rotating FFT owner's protocol and through two lanes (two streams), both
 * required to reproduce that audio bit for bit.
 *
 *   gcc -O2 -Iinclude examples/c_host.c -Lradio-core_amd/radiocore/_lib -lrcfm -lm \
 *       -Wl,-rpath,$PWD/radio-core_amd/radiocore/_lib -o examples/c_host
 *   examples/c_host out_dir          (needs an MI355X)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rcfm.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != RCFM_OK) {                                                            \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, rcfm_last_error());            \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

enum { N = 600000, C = 3, B = 60000, A = 12000, BUFFERS = 3 };

/* three FM stations (stereo multiplex with a 19 kHz pilot) on their channel centres, built in the time domain */
static void synth(float* x /* [N][2] */, int buffer, const double* offset_hz) {
    static double phase[C];
    if (buffer == 0) memset(phase, 0, sizeof(phase));
    memset(x, 0, sizeof(float) * 2 * N);
    const double two_pi = 6.283185307179586476925;
    for (int c = 0; c < C; ++c) {
        double ph = phase[c];
        for (int n = 0; n < N; ++n) {
            const double t = (double)(n + (double)buffer * N) / N;
            const double l = 0.3 * sin(two_pi * (400 + 130 * c) * t), r = 0.3 * sin(two_pi * (1000 + 70 * c) * t + 1.0);
            const double mpx = 0.3 * (l + r) + 0.1 * sin(two_pi * 19000 * t) + 0.3 * (l - r) * sin(two_pi * 38000 * t);
            ph += two_pi * 18750.0 * mpx / N;                     /* 75 kHz deviation scaled to the 60 kHz channel */
            const double arg = ph + two_pi * offset_hz[c] * t;
            x[2 * n] += (float)(0.3 * cos(arg));
            x[2 * n + 1] += (float)(0.3 * sin(arg));
        }
        phase[c] = ph;
    }
}

int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : ".";
    int devices = 0;
    CHECK(rcfm_device_count(&devices));
    if (devices < 1) {
        fprintf(stderr, "no HIP device\n");
        return 2;
    }
    /* channel geometry exactly as the Python Tuner computes it (tuner.py:163-174): centres 100.00 / 100.05 / 99.90 MHz,
     * 60 kHz channels, input bandwidth requested as N = 600 000 */
    const double centre[C] = {100.00e6, 100.05e6, 99.90e6};
    const double f_in = ((99.90e6 - B / 2.0) + (100.05e6 + B / 2.0)) / 2.0;
    int64_t roll[C];
    int32_t bw[C];
    double offset[C];
    for (int c = 0; c < C; ++c) {
        roll[c] = (int64_t)(f_in - centre[c]);
        bw[c] = B;
        offset[c] = centre[c] - f_in;
    }
    rcfm_tuner_t tuner;
    rcfm_demod_t demod;
    rcfm_feeder_t feeder;
    rcfm_comm_t comm;
    CHECK(rcfm_tuner_create(N, C, roll, bw, &tuner));
    CHECK(rcfm_tuner_shard(tuner, 0, C));
    CHECK(rcfm_demod_create(RCFM_WBFM, C, B, A, 75e-6, 0, &demod));
    CHECK(rcfm_feeder_create((size_t)N * 8, 2, NULL, &feeder));
    unsigned char token[RCFM_UNIQUE_ID_BYTES];
    CHECK(rcfm_comm_unique_id(token));
    CHECK(rcfm_comm_init_rank(1, 0, token, &comm));

    float* ring = (float*)malloc(sizeof(float) * 2 * N * BUFFERS);
    float* audio_host = (float*)malloc(sizeof(float) * C * A * 2);
    void *block = NULL, *gathered = NULL;
    if (!ring || !audio_host) return 3;
    CHECK(rcfm_host_register(ring, sizeof(float) * 2 * N * BUFFERS));
    CHECK(rcfm_malloc(&block, sizeof(float) * C * A * 2));
    CHECK(rcfm_malloc(&gathered, sizeof(float) * C * A * 2));
    for (int b = 0; b < BUFFERS; ++b) synth(ring + (size_t)b * 2 * N, b, offset);

    char path[1024];
    snprintf(path, sizeof(path), "%s/c_host_input.bin", dir);
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(ring, sizeof(float), (size_t)2 * N * BUFFERS, f) != (size_t)2 * N * BUFFERS) return 4;
    fclose(f);
    snprintf(path, sizeof(path), "%s/c_host_audio.bin", dir);
    f = fopen(path, "wb");
    if (!f) return 4;

    CHECK(rcfm_feeder_submit(feeder, ring));
    for (int b = 0; b < BUFFERS; ++b) {
        if (b + 1 < BUFFERS) CHECK(rcfm_feeder_submit(feeder, ring + (size_t)(b + 1) * 2 * N));
        void* x = NULL;
        CHECK(rcfm_feeder_acquire(feeder, NULL, &x));
        CHECK(rcfm_tuner_load(tuner, x, NULL));
        CHECK(rcfm_pipeline_run(tuner, demod, 0, C, block, NULL));
        CHECK(rcfm_feeder_release(feeder, NULL));
        CHECK(rcfm_gather_audio(comm, 0, block, (size_t)C * A * 2, gathered, NULL));
        CHECK(rcfm_memcpy_d2h(audio_host, gathered, sizeof(float) * C * A * 2, NULL));
        CHECK(rcfm_stream_sync(NULL));
        double peak = 0.0;
        for (size_t i = 0; i < (size_t)C * A * 2; ++i) {
            if (!isfinite(audio_host[i])) {
                fprintf(stderr, "non-finite audio\n");
                return 5;
            }
            if (fabs(audio_host[i]) > peak) peak = fabs(audio_host[i]);
        }
        printf("buffer %d: %d channels x %d stereo samples, peak %.4f\n", b, C, A, peak);
        if (fwrite(audio_host, sizeof(float), (size_t)C * A * 2, f) != (size_t)C * A * 2) return 4;
    }
    fclose(f);

    /* ---- the same buffers through the ROTATING FFT OWNER's protocol on a one-rank communicator ---------------------
     * (rcfm.h, "the rotating FFT owner's hand-over"): the owner transforms buffer b into slot `own`, hands the bins this
     * rank's channels read to the reader's slot `mine` point to point (here: to itself, a device copy inside one
     * ncclGroup), the reader adopts them and runs its channels.  With G GPUs the only differences are peer != 0 and
     * that owner and reader are different processes.  A fresh demodulator (fresh de-emphasis state) must reproduce the
     * audio file written above, bit for bit: the spectrum the channels read is the same. */
    {
        int64_t halo = 0, nn = 0, first_bin = 0, nbins = 0;
        rcfm_demod_t demod2;
        void *own = NULL, *mine = NULL, *xdev = NULL;
        CHECK(rcfm_demod_create(RCFM_WBFM, C, B, A, 75e-6, 0, &demod2));
        CHECK(rcfm_tuner_spectrum_layout(tuner, &halo, &nn));
        CHECK(rcfm_malloc(&own, sizeof(float) * 2 * (size_t)(nn + 2 * halo)));
        CHECK(rcfm_malloc(&mine, sizeof(float) * 2 * (size_t)(nn + 2 * halo)));
        CHECK(rcfm_malloc(&xdev, sizeof(float) * 2 * N));
        CHECK(rcfm_tuner_window(tuner, 0, C, &first_bin, &nbins));
        /* a circular window as at most two pieces [a, b) of [0, n) */
        int64_t seg[2][2] = {{first_bin, first_bin + nbins}, {0, 0}};
        if (nbins >= nn) {
            seg[0][0] = 0;
            seg[0][1] = nn;
        } else if (first_bin + nbins > nn) {
            seg[0][1] = nn;
            seg[1][1] = first_bin + nbins - nn;
        }
        snprintf(path, sizeof(path), "%s/c_host_audio.bin", dir);
        f = fopen(path, "rb");
        float* want = (float*)malloc(sizeof(float) * C * A * 2);
        if (!f || !want) return 4;
        for (int b = 0; b < BUFFERS; ++b) {
            CHECK(rcfm_memcpy_h2d(xdev, ring + (size_t)b * 2 * N, sizeof(float) * 2 * N, NULL));
            CHECK(rcfm_tuner_attach_spectrum(tuner, own, 0, 0));       /* owner: FFT into its slot */
            CHECK(rcfm_tuner_load(tuner, xdev, NULL));
            CHECK(rcfm_comm_group_start(comm));                         /* hand-over: owner -> reader (rank 0 -> rank 0) */
            for (int k = 0; k < 2; ++k) {
                const size_t n = (size_t)(seg[k][1] - seg[k][0]);
                const size_t at = sizeof(float) * 2 * (size_t)(halo + seg[k][0]);
                CHECK(rcfm_send_bins(comm, 0, (const char*)own + at, n, NULL));
                CHECK(rcfm_recv_bins(comm, 0, (char*)mine + at, n, NULL));
            }
            CHECK(rcfm_comm_group_end(comm));
            CHECK(rcfm_tuner_attach_spectrum(tuner, mine, 0, 0));      /* reader: adopt the bins, run its channels */
            CHECK(rcfm_tuner_adopt(tuner, 0, C, NULL));
            CHECK(rcfm_pipeline_run(tuner, demod2, 0, C, block, NULL));
            CHECK(rcfm_memcpy_d2h(audio_host, block, sizeof(float) * C * A * 2, NULL));
            CHECK(rcfm_stream_sync(NULL));
            if (fread(want, sizeof(float), (size_t)C * A * 2, f) != (size_t)C * A * 2) return 4;
            if (memcmp(want, audio_host, sizeof(float) * C * A * 2) != 0) {
                fprintf(stderr, "rotating owner: buffer %d differs from the direct path\n", b);
                return 6;
            }
            printf("rotating owner, buffer %d: window of %lld of %lld bins handed over, audio identical\n", b,
                   (long long)(nbins < nn ? nbins : nn), (long long)nn);
        }
        fclose(f);
        free(want);
        CHECK(rcfm_tuner_attach_spectrum(tuner, NULL, 0, 0));
        CHECK(rcfm_demod_destroy(demod2));
        CHECK(rcfm_free(own));
        CHECK(rcfm_free(mine));
        CHECK(rcfm_free(xdev));
    }
    /* ---- the same buffers through TWO LANES: consecutive buffers on alternating streams ---------------------------------
     * (rcfm.h, RCFM_OPT_STATE_FENCE).  One handle set per lane -- tuner (spectrum + scratch), demodulator (workspaces),
     * device input and output -- and ONE de-emphasis state: lane 1's demodulator is bound to lane 0's, and the fence orders
     * the launches that touch the state across the two streams.  Nothing is waited for until all buffers are queued; the
     * audio must equal the file written by the one-buffer-at-a-time loop above, bit for bit. */
    {
        enum { LANES = 2 };
        rcfm_tuner_t ltuner[LANES];
        rcfm_demod_t ldemod[LANES];
        void *lstream[LANES], *lx[LANES], *lout[BUFFERS];
        for (int k = 0; k < LANES; ++k) {
            CHECK(rcfm_tuner_create(N, C, roll, bw, &ltuner[k]));
            CHECK(rcfm_tuner_shard(ltuner[k], 0, C));
            CHECK(rcfm_demod_create(RCFM_WBFM, C, B, A, 75e-6, 0, &ldemod[k]));
            if (k) CHECK(rcfm_demod_bind_state(ldemod[k], ldemod[0], 0, 0, NULL));
            CHECK(rcfm_stream_create(&lstream[k]));
            CHECK(rcfm_malloc(&lx[k], sizeof(float) * 2 * N));
        }
        CHECK(rcfm_demod_set_option(ldemod[0], RCFM_OPT_STATE_FENCE, 1));     /* after the binding */
        CHECK(rcfm_stream_sync(NULL));
        for (int b = 0; b < BUFFERS; ++b) {
            const int k = b % LANES;
            CHECK(rcfm_malloc(&lout[b], sizeof(float) * C * A * 2));
            /* stream-ordered on lane k: its previous buffer's kernels have read lx[k] before this copy overwrites it */
            CHECK(rcfm_memcpy_h2d(lx[k], ring + (size_t)b * 2 * N, sizeof(float) * 2 * N, lstream[k]));
            CHECK(rcfm_tuner_load(ltuner[k], lx[k], lstream[k]));
            CHECK(rcfm_pipeline_run(ltuner[k], ldemod[k], 0, C, lout[b], lstream[k]));
        }
        for (int k = 0; k < LANES; ++k) CHECK(rcfm_stream_sync(lstream[k]));
        snprintf(path, sizeof(path), "%s/c_host_audio.bin", dir);
        f = fopen(path, "rb");
        float* want = (float*)malloc(sizeof(float) * C * A * 2);
        if (!f || !want) return 4;
        for (int b = 0; b < BUFFERS; ++b) {
            CHECK(rcfm_memcpy_d2h(audio_host, lout[b], sizeof(float) * C * A * 2, NULL));
            CHECK(rcfm_stream_sync(NULL));
            if (fread(want, sizeof(float), (size_t)C * A * 2, f) != (size_t)C * A * 2) return 4;
            if (memcmp(want, audio_host, sizeof(float) * C * A * 2) != 0) {
                fprintf(stderr, "two lanes: buffer %d differs from the one-buffer-at-a-time loop\n", b);
                return 7;
            }
            printf("two lanes, buffer %d on stream %d: audio identical\n", b, b % LANES);
            CHECK(rcfm_free(lout[b]));
        }
        fclose(f);
        free(want);
        for (int k = 0; k < LANES; ++k) {
            CHECK(rcfm_demod_destroy(ldemod[k]));
            CHECK(rcfm_tuner_destroy(ltuner[k]));
            CHECK(rcfm_stream_destroy(lstream[k]));
            CHECK(rcfm_free(lx[k]));
        }
    }
    CHECK(rcfm_host_unregister(ring));
    CHECK(rcfm_comm_destroy(comm));
    CHECK(rcfm_feeder_destroy(feeder));
    CHECK(rcfm_demod_destroy(demod));
    CHECK(rcfm_tuner_destroy(tuner));
    CHECK(rcfm_free(block));
    CHECK(rcfm_free(gathered));
    free(ring);
    free(audio_host);
    printf("ok\n");
    return 0;
}
