// twv_xcd.hpp -- host-visible interface of the XCD-per-stream generation kernel (twv_wavenet_xcd.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "twv_layout.hpp"

namespace twv {

constexpr int kXcdMaxLayers = 31;      // chain workgroup: wave 0 holds the causal layer + 3 layers, waves 1..7 four layers each
constexpr int kXcdStreams = 32;        // up to four streams per XCD (stream b runs on XCD b % 8)
constexpr int kXcdLcRing = 16;         // steps of lc projections the lc workgroups may run ahead of the chain
constexpr int kXcdXlFloats = 13 * 64 * 4;   // per layer: the chain's register image [13 float4][64 lanes]
constexpr int kXcdXcFloats = 8 * 64 * 4;    // causal kernel in the chain's lane order

// conditioning buffer of the XCD path: [64 ints header][B][NL][64] gc projections, then the rows the lc workgroups read
enum { XH_MAGIC = 0, XH_MODE = 1, XH_ROWS = 2, XH_WORDS = 64 };
constexpr int kXcdCondMagic = 0x58434431;
enum { XLC_NONE = 0, XLC_UPSAMPLED = 1, XLC_MEL = 2 };

// exchange area of one stream, in 8-byte granules {tag, value}
struct XcdExch {
    static constexpr long long ZX = 0;                                  // [32][64] x {z, tag | layer input, tag}: chain -> skip | service
    static constexpr long long PG = ZX + 32 * 128;                      // [32][64] tap-0 chunk                 service -> chain
    static constexpr long long LG = PG + 32 * 64;                       // [32][64] lc projection               service -> chain
    static constexpr long long H1 = LG + 32 * 64;                       // [512] relu(skip sum)                 skip -> conv1
    static constexpr long long PT = H1 + 512;                           // [16][32] conv1d_2 chunk partials     conv1 -> sampler
    static constexpr long long LCR = PT + 512;                          // [ring][32][64] lc projections        lc -> service
    static constexpr long long CTRL = LCR + (long long)kXcdLcRing * 32 * 64;   // [64] progress, abort
    static constexpr long long MARK = CTRL + 64;                        // [32 roles][8 waves] {step, stage} markers (instrumented build)
    static constexpr long long WORDS = MARK + 256;
};

struct XcdLaunch {
    const float* P;
    float* state;
    const float* cond;
    const void* first_input;
    const void* forced;
    const float* uniforms;
    float* out;
    int* status;
    float* dbg;
    int dbg_steps;
    unsigned long long* prof;      // optional [prof_steps][64] s_memtime stamps of stream 0 (tuning aid; see scripts/xcd_phase_profile.py)
    int prof_steps;
    int prof_stream;               // the stream whose workgroups stamp (0; TWV_XCD_PROF_STREAM picks another one: tuning aid)
    int B, T;
    unsigned long long* exch;      // [B][XcdExch::WORDS]
    int* roles;                    // [8] role tickets per XCD (zeroed before the launch)
    Layout lay;
};

bool xcd_model_ok(const Layout& L);                       // shape the kernel is written for
int xcd_lc_workgroups(const Layout& L);                   // lc workgroups per stream
int xcd_workgroups_per_stream(const Layout& L);
size_t xcd_exchange_bytes(int batch);                     // exchange area + role tickets
int xcd_launch(const XcdLaunch& a, hipStream_t st);       // TWV_OK or an error code (text via twv_fail)
void xcd_pack(float* packed, const float* blob, const Layout& L, hipStream_t st);

}  // namespace twv
