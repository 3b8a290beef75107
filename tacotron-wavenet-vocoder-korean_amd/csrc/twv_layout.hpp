// twv_layout.hpp -- streaming ("packed") weight layout, per-stream state layout and conditioning layout of the
// WaveNet generation path.  Plain-old-data shared by the host-side C-ABI and the gfx950 kernels.
//
// Unit of the packed layout is the TILE: 64 outputs x 32 reduction terms, stored [kq=8][lane=64][4] floats
// (8 KiB) so that one wave reads it with eight coalesced 16-byte-per-lane loads and every lane ends up holding
// the 32 weights of ITS output in registers.  A tile's 32 terms are exactly one chunk of the arithmetic
// contract's chunked dot product.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace twv {

constexpr int kTile = 2048;          // floats per tile
constexpr int kMaxLayers = 64;

// per-layer block of the packed layout (offsets in floats from the layer base).  [T0 .. BD] is the CHAIN block: it is
// contiguous so that the loader waves stage it into an LDS slot with 21 one-KiB LDS-DMA pieces.
struct LayerOff {
    static constexpr int T0 = 0;             // conv_filter|conv_gate kernel, tap 0 (x[t-d])   model.py:68-69
    static constexpr int T1 = kTile;         // conv_filter|conv_gate kernel, tap 1 (x[t])
    static constexpr int WD = 2 * kTile;     // dense kernel, HALF tile [kq=8][32][4]           model.py:89
    static constexpr int BFG = 2 * kTile + 1024;   // 64: conv_filter bias | conv_gate bias
    static constexpr int BD = BFG + 64;      // 32: dense bias (+32 pad)
    static constexpr int SK = BD + 64;       // NSJ tiles: skip kernel   model.py:96
    static constexpr int CHAIN_FLOATS = SK;  // 5248
    // BS = SK + NSJ*kTile : S floats skip bias
};
// LDS slot of one (layer, step) item staged by the loader waves: the layer's [T1|WD|BFG|BD] block (13 one-KiB LDS-DMA
// pieces, same offsets as LayerOff minus T1), then x[t-d], the lc projection row and the loader-computed tap-0 chunk.
struct SlotOff {
    static constexpr int PIECES = 13;                      // ceil((CHAIN_FLOATS - kTile) * 4 / 1024)
    static constexpr int T1 = 0;
    static constexpr int WD = LayerOff::WD - LayerOff::T1;     // 2048
    static constexpr int BFG = LayerOff::BFG - LayerOff::T1;   // 3072
    static constexpr int BD = LayerOff::BD - LayerOff::T1;     // 3136
    static constexpr int XO = PIECES * 256;                // 3328: x[t-d] (32, duplicated to 64)
    static constexpr int LC = XO + 64;                     // lc projection row of this layer (64)
    static constexpr int PK = LC + 64;                     // per lane {tap-0 chunk, conv bias, gc projection, lc projection}: one b128
    static constexpr int FLOATS = PK + 256;                // 3712 floats = 14848 B
};

struct Layout {
    // model
    int NL, S, Q, O, Opad, scalar, ifw, use_bias, G, gc_card, L, n_up;
    int up[4];
    int NSJ;   // S/64   skip / post1 output blocks
    int NCH;   // S/32   chunks of the post 1x1 convs
    int NOJ;   // Opad/64 output blocks of the last conv
    int NCA;   // ceil(ifw/32) causal chunks (scalar input)
    int NLC;   // ceil(L/32)
    int NGC;   // ceil(G/32)
    int nr_mix;
    int nslot;  // LDS slots of the chain-weight ring
    // packed offsets (floats)
    long long off_meta;      // int32: dil[64], ring_off[64] (ring offsets in floats within a stream's ring area)
    long long off_causal;    // scalar: NCA tiles (outputs duplicated); one-hot: [2][Q][32]
    long long off_layer0, layer_stride;
    long long off_w1, off_b1, off_w2, off_b2;
    long long off_lcw, lcw_stride;   // per layer NLC tiles: lc_filter|lc_gate
    long long off_gcw, gcw_stride;   // per layer NGC tiles: gc_filter|gc_gate
    long long off_gcemb;             // [card][G]
    long long off_up[4];             // [f][2]
    long long off_xl, off_xc;        // XCD path (twv_wavenet_xcd.hip): per layer the chain's register image; causal kernel in its lane order (0 = absent)
    long long packed_floats;
    // canonical blob offsets (floats): TF checkpoint tensors, order of DESIGN.md
    long long c_causal, c_gcemb, c_layer0, c_layer_stride;
    long long c_wf, c_bf, c_wg, c_bg, c_gcf, c_gcg, c_lcf, c_lcg, c_wd, c_bd, c_ws, c_bs;  // within a layer
    long long c_w1, c_b1, c_w2, c_b2, c_up[4], blob_floats;
    // per-stream state (floats): [hist 64][meta 64 ints][ringpos 64 ints][lcprev NL*64][rings sum(d)*32]
    long long st_hist, st_meta, st_ringpos, st_lcprev, st_ring, state_stride;
    int ring_floats;
};

// state meta words
enum { M_TABS = 0, M_HPOS = 1, M_PREV_VALID = 2, M_QPREV = 3 };

}  // namespace twv
