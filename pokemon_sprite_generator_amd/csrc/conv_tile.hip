// One (dtype, BM x BN) instantiation of the implicit-GEMM kernel: compiled once per tile by the Makefile with
// -DPSG_TILE_BF16=0|1 -DPSG_TILE_BM=.. -DPSG_TILE_BN=..
#include "conv_gemm_kernel.h"
#ifndef PSG_TILE_BM          // (a bare `hipcc -c conv_tile.hip` still compiles: the smallest fp32 tile)
#define PSG_TILE_BF16 0
#define PSG_TILE_BM 64
#define PSG_TILE_BN 64
#endif

namespace psg {
#if PSG_TILE_BF16
typedef bf16_t tile_t;
#else
typedef float tile_t;
#endif
template int launch_conv<tile_t, PSG_TILE_BM, PSG_TILE_BN>(const ConvP&, hipStream_t);
template int set_conv_attrs<tile_t, PSG_TILE_BM, PSG_TILE_BN>();
}  // namespace psg
