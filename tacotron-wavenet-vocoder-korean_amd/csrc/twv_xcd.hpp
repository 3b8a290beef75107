// twv_xcd.hpp -- host-visible interface of the XCD-per-stream generation kernel (twv_wavenet_xcd.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "twv_layout.hpp"

namespace twv {

constexpr int kXcdSeg0Layers = 30;     // first chain workgroup: waves 0..5 four layers each, waves 6 and 7 three (wave 7 also runs the causal layer + sampler)
constexpr int kXcdMaxLayers = 50;      // a second chain workgroup takes layers 30.. (hparams.py has 50); limit: LDS of the service workgroup (tiles of layers 0 .. NL-33) and of the skip workgroups (value slots of two streams + tiles of layers 0 .. NL-41)
constexpr int kXcdLs = 64;             // layer slots of the per-layer exchange arrays
constexpr int kXcdStreams = 32;        // up to four streams per XCD (stream b runs on XCD b % 8), each with its own chain workgroup
constexpr int kXcdManyFrom = 21;       // batch from which the many-streams kernel is the default choice (measured: B = 16 9.25 vs 9.57 us per step, B = 24 9.65 vs 9.54, B = 32 11.0 vs 9.60)
constexpr int kXcdManyStreams = 96;    // the many-streams kernel (30 layers or fewer): up to twelve per XCD, two per chain workgroup
constexpr int kXcdLcRing = 16;         // steps of lc projections the lc workgroups may run ahead of the chain
#ifndef TWV_MANY_LC_RING
#define TWV_MANY_LC_RING 8
#endif
constexpr int kXcdManyLcRing = TWV_MANY_LC_RING;   // the same for the many-streams kernel: the first 8 slots of the same area (>= 8: its lc role is throttled every fourth row).
                                                   // 16 -> 8: 1.03 -> 0.37 MB written per step at batch 64 (the ring of an XCD's streams no longer spills its L2), step time equal
constexpr int kXcdXlFloats = 13 * 64 * 4;   // per layer: the chain's register image [13 float4][64 lanes]
constexpr int kXcdXcFloats = 8 * 64 * 4;    // causal kernel in the chain's lane order

// conditioning buffer of the XCD path: [64 ints header][B][NL][64] gc projections, then the rows the lc workgroups read
enum { XH_MAGIC = 0, XH_MODE = 1, XH_ROWS = 2, XH_WORDS = 64 };
constexpr int kXcdCondMagic = 0x58434431;
enum { XLC_NONE = 0, XLC_UPSAMPLED = 1, XLC_MEL = 2 };

// exchange area of one stream, in 8-byte granules {tag, value}
struct XcdExch {
    static constexpr long long ZX = 0;                                  // [Ls][64] x {z, tag | layer input, tag}: chain -> skip | service
    static constexpr long long PG = ZX + kXcdLs * 128;                  // [Ls][64] addend ((tap-0 chunk + bias) + gc) + lc (AC-1b)   service -> chain
    static constexpr long long LG = PG + kXcdLs * 64;                   // [Ls][64] (unused since round 5: the lc projection travels inside the addend)
    static constexpr long long H1 = LG + kXcdLs * 64;                   // [512] relu(skip sum)                 skip -> conv1
    static constexpr long long PT = H1 + 512;                           // [16][32] conv1d_2 chunk partials     conv1 -> sampler
    static constexpr long long LCR = PT + 512;                          // [ring][Ls][64] lc projections        lc -> service
    static constexpr long long CTRL = LCR + (long long)kXcdLcRing * kXcdLs * 64;   // [64] progress, abort
    static constexpr long long MARK = CTRL + 64;                        // [32 roles][8 waves] {step, stage} markers (instrumented build)
    static constexpr long long SEG = MARK + 256;                        // [64] residual vector, first -> second chain workgroup (more than 30 layers)
    static constexpr long long DONE = SEG + 64;                         // [64] end of a teacher-forced step, second chain workgroup -> head
    static constexpr long long SKT = DONE + 64;                         // [3 hops][2 halves][2 pairs][64 lanes][2] running skip total, layer group -> next group (many-streams kernel)
    static constexpr long long H2 = SKT + 3 * 2 * 256;                  // [512] relu(conv1d_1)                conv1 -> conv1 (one-hot model: conv1d_2 is split by OUTPUT)
    static constexpr long long QL = H2 + 512;                           // [64 lanes][4] logits of class lane + 64 k   conv1 -> sampler (one-hot model)
    static constexpr long long WORDS = QL + 256;
    // (slots of layers a model does not have are never touched: they cost address space, not cache)
};

struct XcdLaunch {
    const float* P;
    float* state;
    const float* cond;
    const void* first_input;
    const void* forced;
    const float* uniforms;         // scalar input: (B, T, nr_mix + 1) float; one-hot: (B, T) double
    float temperature;             // one-hot model: generate.py:220
    float* out;                    // scalar input: float samples; one-hot: int32 class ids
    int* status;
    float* dbg;
    int dbg_steps;
    unsigned long long* prof;      // optional [prof_steps][64] s_memtime stamps of stream 0 (tuning aid; see scripts/xcd_phase_profile.py)
    int prof_steps;
    int prof_stream;               // the stream whose workgroups stamp (0; TWV_XCD_PROF_STREAM picks another one: tuning aid)
    int B, T;
    int many;                      // option "xcd_many": 0 = from batch kXcdManyFrom on, 1 = at every batch, 2 = only above batch 32
    unsigned long long* exch;      // [B][XcdExch::WORDS]
    int* roles;                    // [8] role tickets per XCD (zeroed before the launch)
    Layout lay;
};

bool xcd_model_ok(const Layout& L);                       // shape the kernel is written for
int xcd_max_streams(const Layout& L);                     // MoL: 96 (30 layers or fewer), 16 above; one-hot: 32 / 16
bool xcd_uses_many(const Layout& L, int batch, int many_opt);   // wn_xcd_many_kernel rather than wn_xcd_generate_kernel for this batch
int xcd_lc_workgroups(const Layout& L);                   // lc workgroups per stream
int xcd_workgroups_per_stream(const Layout& L);
size_t xcd_exchange_bytes(int batch);                     // exchange area + role tickets
int xcd_launch(const XcdLaunch& a, hipStream_t st);       // TWV_OK or an error code (text via twv_fail)
void xcd_pack(float* packed, const float* blob, const Layout& L, hipStream_t st);

}  // namespace twv
