// Two builds of the tile kernels from one source.
//
// A tile is W adjacent lines x L points.  W = 16 (one 128-byte segment per row, 256..1024 threads) is what streams
// best when a launch has hundreds of tiles per CU-round -- every batched configuration.  A SINGLE channel
// (BASELINE configs[1]: one WBFM.run per call, tests/benchmark.py:29-31,85 in the reference) is 30 such tiles on 256
// CUs and every launch lasts one tile's latency with two waves per SIMD; with W = 8 the same channel is 60 tiles of
// half the threads -- one wave per SIMD, half the LDS traffic per CU -- and the data is L2-resident anyway, so the
// 64-byte segments cost nothing.  The translation units that hold tile kernels are therefore compiled twice:
// as they are (namespace rcfm, W = 16) and with -DRCFM_TILE_W=8 (namespace rcfm::narrow); api.hip picks per call.
#pragma once

#ifndef RCFM_TILE_W
#define RCFM_TILE_W 16
#endif

#if RCFM_TILE_W == 16
#define RCFM_NS_OPEN
#define RCFM_NS_CLOSE
#define RCFM_NARROW_BUILD 0
#else
#define RCFM_NS_OPEN namespace narrow {
#define RCFM_NS_CLOSE }
#define RCFM_NARROW_BUILD 1
#endif
