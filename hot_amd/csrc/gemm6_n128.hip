// Second build of csrc/gemm6.hip with 256 x 128 tiles (8 waves as 4 x 2, a wave owns 64 x 64): the column tail of a
// product whose width is not a multiple of 256 (fc_v / fc_o of C2: 320 columns).  Exports hotk_gemm6_nn_cols_n128 only.
#undef HOTK_G6_BM
#undef HOTK_G6_BN
#undef HOTK_G6_WM
#undef HOTK_G6_WN
#undef HOTK_G6_WGS
#define HOTK_G6_BM 256
#define HOTK_G6_BN 128
#define HOTK_G6_WM 4
#define HOTK_G6_WN 2
#define HOTK_G6_WGS 1
#define HOTK_G6_VARIANT _n128
#include "gemm6.hip"
