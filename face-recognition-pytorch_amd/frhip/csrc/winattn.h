// Shared by the two window-attention implementations: winattn.hip (VALU, fp32-exact, any dtype) and
// winattn_mfma.hip (bf16 MFMA tiles).
#pragma once
#include "common.h"

namespace frhip {

constexpr int WA_N = 49, WA_D = 32;     // largest window (7x7 tokens); smaller windows (6x6, 3x3) use the same tiles

// Window geometry: `ws` x `ws` tokens, optional cyclic shift (SW-MSA, nets/AlterNet_SwinV2_FAN.py:420-440): token
// (ty,tx) of window (wy,wx) of the ROLLED image is pixel ((wy*ws+ty+shift) % H, (wx*ws+tx+shift) % W) of the original,
// and the output goes back to that same pixel (the reverse roll).  `region` is the 3x3 region id of the rolled
// position used by the reference's attention mask (:375-397): tokens of different regions get -100 added.
struct WaGeom { int H, W, ws, shift, n; };

__device__ __forceinline__ size_t wa_pixel(int win, int tok, const WaGeom& g, int* region) {
    const int wpr = g.W / g.ws, wpi = (g.H / g.ws) * wpr;
    const int b = win / wpi, r = win - b * wpi, wy = r / wpr, wx = r - wy * wpr;
    const int ty = tok / g.ws, tx = tok - ty * g.ws;
    const int hs = wy * g.ws + ty, wsx = wx * g.ws + tx;                 // position in the rolled image
    int hh = hs + g.shift, ww = wsx + g.shift;
    if (hh >= g.H) hh -= g.H;
    if (ww >= g.W) ww -= g.W;
    if (region) {
        const int rh = hs < g.H - g.ws ? 0 : (hs < g.H - g.shift ? 1 : 2);
        const int rw = wsx < g.W - g.ws ? 0 : (wsx < g.W - g.shift ? 1 : 2);
        *region = g.shift > 0 ? rh * 3 + rw : 0;
    }
    return ((size_t)b * g.H + hh) * g.W + ww;
}

// destinations (fp32 [C] each, any may be null) of the column sums of the q / k / v thirds of dqkv
struct WaColsum { float* p[3]; };

// bf16 MFMA implementation (winattn_mfma.hip); same operands as frhip_winattn_fwd / _bwd
int winattn_mfma_fwd(const void* qkv, const float* bias, const float* scale, void* out, int nwin, const WaGeom& g, int C,
                     int heads, hipStream_t stream);
int winattn_mfma_bwd(const void* qkv, const void* dout, const float* bias, const float* scale, void* dqkv, float* dbias,
                     float* dscale, const WaColsum& colsum, int nwin, const WaGeom& g, int C, int heads, hipStream_t stream);

}  // namespace frhip
