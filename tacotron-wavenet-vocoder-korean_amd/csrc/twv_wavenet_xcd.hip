// twv_wavenet_xcd.hip -- the XCD-per-stream WaveNet generation kernel (MI355X: 8 XCDs x 32 CUs, 4 MiB L2 and 16 MB of
// vector registers per XCD).
//
// Replaces, for the hparams-default MoL vocoder (wavenet/model.py:41-167,215-245, wavenet/mixture.py:84-114, generate.py:199-233;
// scalar input, initial_filter_width 32, R = D = 32, S = 512, out_channels <= 32, up to 50 layers, batch <= 32 (16 above 30 layers)):
// the same per-sample chain as wn_generate_kernel (twv_wavenet.hip), but laid out for the chip instead of for one CU:
//
//   * A STREAM LIVES ON ONE XCD (stream b on XCD b % 8).  Every workgroup reads its XCC id and takes a role ticket of that XCD, so all workgroups of a stream
//     share one L2: a hand-off is a PLAIN 8-byte store {tag, value} (the line stays in the XCD's L2) polled with sc1 loads
//     (bypass the reader's L1): 273 ns one way instead of the 590 ns of the agent-scope (write-through + memory-side) form
//     (profiles/r02_xcd_chain_ubench.txt).
//   * EVERY WEIGHT IS REGISTER-RESIDENT for the whole launch, spread over the XCD's CUs: nothing is re-streamed per step.
//       chain workgroup   (1 per stream): 8 waves x 4 layers: tap-1 conv kernel + dense kernel of a layer = 48 VGPRs per lane
//       service workgroup (1 per stream): tap-0 conv kernels, the delay lines (model.py:49-64 queues; in the stream's state buffer)
//       skip workgroups   (8 per XCD): slice g of the 30 skip kernels (model.py:94-96)
//       conv1 workgroups  (8 per XCD): slice g of conv1d_1 and its two chunks of conv1d_2 (model.py:158-165)
//       lc workgroups   (2-4 per XCD): create_upsample + lc_filter/lc_gate projections (model.py:102-111,75-83), running ahead of
//                              the chain through a ring: neither the upsampled condition nor a projection table exists in HBM
//     More than 30 layers (hparams.py's default stack has 50): a SECOND chain workgroup takes layers 30.. (one L2 hop away), the
//     service / skip waves keep the tiles of the early layers in LDS (their values are needed last) and the lc waves hold two layers;
//     this is a kernel instantiation of its own (BIGK), the 30-layer kernel's code is untouched by it: 14.7 us/step at 50 layers
//     against 32.9 on the generic kernel.
//     The skip / conv1 / lc workgroups hold weights only, so with more than 8 streams ONE set per XCD serves the XCD's streams in
//     turn (2 ns + 16 + n_lc <= 28 of the 32 CUs for ns = 4): the streams settle a fraction of a microsecond apart, B = 16 keeps
//     the single-stream step time (9.7 us), B = 32 runs at 10.8 us (2.97 M samples/s).
//   * THE CHAIN IS A RELAY OF EIGHT WAVES.  A layer is 32 v_fmac_f32_dpp (row_newbcast feeds x[k] to the fma: no v_readlane,
//     no LDS operand reads) -> bias/conditioning adds -> rational tanh/sigmoid -> v_permlane32_swap -> 16 v_fmac_f32_dpp +
//     v_permlane16_swap for the dense 1x1 (twv_dpp.hpp): 243 ns per layer in the product (200 for the arithmetic alone) against 654 ns
//     in wn_generate_kernel.  A wave hands the residual vector to the next one through a tagged LDS granule (90 ns); nothing but
//     the poll's exit sits between its arrival and the first layer (every branch there costs ~60 cycles: DESIGN.md section 4).  What is not on the sample-to-sample dependency chain
//     (tap-0 chunks, which only need x[t-d]; lc projections; delay-line traffic) never touches the chain workgroup.
//
// Arithmetic: the contract of DESIGN.md (AC-1..AC-4); every dot product is the same fma chains in the same order as in
// wn_generate_kernel and the CPU checker, so the two kernels produce identical bits and share the state layout (G = 1).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <type_traits>
#include "../../include/twv_amd.h"
#include "twv_layout.hpp"
#include "twv_math.hpp"
#include "twv_dev.hpp"
#include "twv_dpp.hpp"
#include "twv_categorical.hpp"
#include "twv_xcd.hpp"

using namespace twv;

#ifndef TWV_CHAIN_PAD
#define TWV_CHAIN_PAD 0
#endif
#ifndef TWV_WATCHDOG_LOG2
#define TWV_WATCHDOG_LOG2 21            // polls before a wait gives up (a tuning build may lower it: -DTWV_WATCHDOG_LOG2=14 finds a deadlock in milliseconds)
#endif
#define TWV_STR2(x) #x
#define TWV_STR(x) TWV_STR2(x)
namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
#define LDSU64(i) (((__attribute__((address_space(3))) volatile unsigned long long*)lds)[(i)])

// Granule traffic goes through ONE buffer descriptor over the stream's exchange area: per access one 32-bit lane offset (VGPR) and
// one uniform offset (SGPR / immediate) -- 64-bit flat addresses of every granule array would not fit next to the weights.
//   load : sc1 (aux bit 4): misses the CU's L1, served by the XCD's L2 (MI355X_MICROARCH: "sc1 loads bypass L1 only")
//   store: plain: write-through the L1, the line STAYS in the XCD's L2 (an sc1 store would drop it to the memory side)
// aux bit 31 = volatile for the compiler (a poll must not be hoisted out of its loop; it also sets sc0, which changes nothing for a
// load).  Stores are NOT volatile (that would make them sc0 sc1 = write-through to the memory side); a compiler barrier pins them.
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
constexpr int kAuxLoad = (int)(16u | 0x80000000u), kAuxStore = 0;
#ifndef TWV_XCD_KF
#define TWV_XCD_KF 1
#endif

__device__ __forceinline__ unsigned long long xb_load(rsrc_t rs, int uword, int lword)
{
    asm volatile("" ::: "memory");      // a poll must be re-issued on every trip of its loop: see below
    const u32x2v v = __builtin_amdgcn_raw_buffer_load_b64(rs, lword * 8, uword * 8, kAuxLoad);
    return ((unsigned long long)v.y << 32) | (unsigned long long)v.x;
}
// two adjacent granules with one load (each half carries its own tag: the halves may come from different stores)
__device__ __forceinline__ u32x4s xb_load2(rsrc_t rs, int uword, int lpair)
{
    asm volatile("" ::: "memory");
    return __builtin_amdgcn_raw_buffer_load_b128(rs, lpair * 16, uword * 8, kAuxLoad);
}
// Every granule load sits behind a compiler barrier.  Aux bit 31 marks the MACHINE instruction volatile, but the IR call is a read-only
// intrinsic, and loop-invariant code motion has twice hoisted a poll out of its loop, which then spins on a register: round 3 in
// wn_xcd_many_kernel (the sampler's four table loads; until the watchdog), round 5 in one instantiation of the one-hot head's logits poll
// (docs/experiments: the abort-flag load of the watchdog went with it -- a hang).  Whether it happens depends on what shares the
// function, i.e. any edit can flip it, so since round 5 the barrier is unconditional (rounds 2-4 kept the batch <= 32 kernel's polls
// plain because "the barrier shifts its schedule": measured now, interleaved A/B, 8.571-8.577 us per step with it against 8.581-8.590
// without).  A load cannot move across the barrier; it costs no instruction.  xb_load_t<BAR> / xbm_load* are the older names.
template <bool BAR> __device__ __forceinline__ unsigned long long xb_load_t(rsrc_t rs, int uword, int lword)
{
    if (BAR) asm volatile("" ::: "memory");
    return xb_load(rs, uword, lword);
}
__device__ __forceinline__ unsigned long long xbm_load(rsrc_t rs, int uword, int lword) { return xb_load_t<true>(rs, uword, lword); }
__device__ __forceinline__ u32x4s xbm_load2(rsrc_t rs, int uword, int lpair)
{
    asm volatile("" ::: "memory");
    return xb_load2(rs, uword, lpair);
}
__device__ __forceinline__ void xb_store(rsrc_t rs, int uword, int lword, unsigned tag, float v)
{
    __builtin_amdgcn_raw_buffer_store_b64(u32x2v{__float_as_uint(v), tag}, rs, lword * 8, uword * 8, kAuxStore);
    asm volatile("" ::: "memory");
}
// two self-tagged 8-byte granules {a, tag}, {b, tag} with ONE 16-byte store (readers load the halves separately)
__device__ __forceinline__ void xb_store2(rsrc_t rs, int uword, int lane, unsigned tag, float va, float vb)
{
    const u32x4s d = {__float_as_uint(va), tag, __float_as_uint(vb), tag};
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, lane * 16, uword * 8, kAuxStore);
    // Store-data hazard: a VMEM store of more than 64 bits reads its data registers over the following cycles, and the compiler's
    // hazard recogniser exempts buffer stores whose soffset is an SGPR -- on gfx950 the very next v_mov into the tuple still tore the
    // granule (lanes 12-15 of every row stored the NEW register contents: tags went missing once in ~10^4 stores).  Keeping the
    // tuple alive as an operand of the wait states makes the window safe whatever the scheduler does.
    asm volatile("s_nop 1" ::"v"(d) : "memory");
}
__device__ __forceinline__ unsigned g_tag(unsigned long long q) { return (unsigned)(q >> 32); }
__device__ __forceinline__ float g_val(unsigned long long q) { return __uint_as_float((unsigned)q); }

// bounded polling with a stream-wide abort word: a wait that runs out raises a watchdog code instead of hanging
struct Poll {
    rsrc_t rs;
    int* status;
    int it;
    bool dead;
};
template <bool BAR = false>
__device__ __forceinline__ bool poll_tick(Poll& p, int code)
{
    if (p.dead) return false;
    if (((++p.it) & 63) == 0) {
        if (xb_load_t<BAR>(p.rs, (int)XcdExch::CTRL + 1, 0) != 0ull) { p.dead = true; return false; }
        if (p.it > (1 << TWV_WATCHDOG_LOG2)) {
            atomicMax(p.status, code);
            xb_store(p.rs, (int)XcdExch::CTRL + 1, 0, 1u, 0.0f);
            p.dead = true;
            return false;
        }
    }
    return true;
}
__device__ __forceinline__ void nap_until(unsigned long long target)
{
#pragma nounroll
    for (int i = 0; i < 4096 && __builtin_amdgcn_s_memtime() < target; ++i) __builtin_amdgcn_s_sleep(8);
}

// chain lane that holds z[k] / x[j] in the granule arrays (twv_dpp.hpp layouts)
__device__ __forceinline__ int lane_of_z(int k) { return ((k & 2) ? 16 : 0) + 2 * (k >> 2) + (k & 1); }
__device__ __forceinline__ int lane_of_x(int j) { return j < 16 ? j : 16 + j; }

// a chain wave's per-layer registers (the dense kernel lives in LDS): tap-1 conv kernel, dense bias as the start value of the dense
// chunk's chain 0 (AC-1b, twv_dpp.hpp: dense_bias_init).  Conv bias, gc and lc projections reach the chain inside the ADDEND the
// service workgroup publishes.
struct ChainRegs { float wc[32]; float bd; };

// phase stamps (instrumented build only): lane 0 of stream 0's workgroups; s_memrealtime (100 MHz, ONE clock for the chip --
// s_memtime counters of different CUs are offset against each other by milliseconds)
#define XSTAMP(cond_, slot_) do { if ((INSTR & 1) && a.prof != nullptr && b == a.prof_stream && t < a.prof_steps && lane == 0 && (cond_)) a.prof[(long long)t * 64 + (slot_)] = wall_clock64(); } while (0)

// wait accounting (INSTR & 4, the many-streams kernel's tuning build, scripts/many_profile.py): every wave adds up the s_memtime ticks it
// spends inside its polls, by kind; at the end lane 0 of XCD 0's waves writes {total, waits...} to prof[(role * 8 + wave) * 8 ..]
#define WACC_DECL() unsigned long long wacc_[6] = {0, 0, 0, 0, 0, 0}, w0_ = 0, wt0_ = 0; if (INSTR & 4) wt0_ = __builtin_amdgcn_s_memtime()
#define WACC_T0() do { if (INSTR & 4) w0_ = __builtin_amdgcn_s_memtime(); } while (0)
#define WACC_T1(i_) do { if (INSTR & 4) wacc_[i_] += __builtin_amdgcn_s_memtime() - w0_; } while (0)
// one step's timeline (the same build): chip-wide 100 MHz clock of a few events of XCD 0 / stream 0 at step kTraceStep -> prof[2048 + slot]
constexpr int kTraceStep = 2000;
#define WTRACE(cond_, slot_) do { if ((INSTR & 4) && a.prof != nullptr && t == kTraceStep && (cond_) && lane == 0) a.prof[2048 + (slot_)] = wall_clock64(); } while (0)
#define WACC_OUT(prof_, slot_, wave_) do { if ((INSTR & 4) && (prof_) != nullptr && (slot_) >= 0 && lane == 0) {                 \
        unsigned long long* o_ = (prof_) + (((slot_) * 8 + (wave_)) * 8);                                                        \
        o_[0] = __builtin_amdgcn_s_memtime() - wt0_;                                                                             \
        o_[1] = wacc_[0]; o_[2] = wacc_[1]; o_[3] = wacc_[2]; o_[4] = wacc_[3]; o_[5] = wacc_[4]; o_[6] = wacc_[5]; } } while (0)
// where a wave is (instrumented build only): read back from the exchange area after a watchdog abort
#define XMARK(role_, stage_) do { if ((INSTR & 1) && lane == 0) xb_store(rs, (int)XcdExch::MARK + (role_) * 8 + (int)(threadIdx.x >> 6), 0, (unsigned)t + 1u, (float)(stage_)); } while (0)

enum { ROLE_CHAIN = 0, ROLE_SERVICE = 1, ROLE_SKIP0 = 2, ROLE_CONV0 = 10, ROLE_LC0 = 18 };     // slots of the stage markers (XMARK)
constexpr int kXcdStreamsPerXcd = kXcdStreams / 8;

struct XArgs {
    XcdLaunch p;
    int n_lc_wg, lc_lpw;      // lc workgroups per XCD, layers per lc wave
    int total_roles;          // role workgroups of the whole launch (all XCDs): what roles_resident counts up to
};

// the streams of one XCD (stream b runs on XCD b % 8): each has its own chain and service workgroup; the skip, conv1 and lc
// workgroups hold weights only, so ONE set per XCD serves them all, stream after stream in a fixed order -- the streams settle a
// fraction of a microsecond apart and every one keeps the single-stream step time
template <int NS> struct XStreams { rsrc_t rs[NS]; int b[NS]; };
__device__ __forceinline__ rsrc_t exch_rsrc(const XcdLaunch& a, int b)
{
    return __builtin_amdgcn_make_buffer_rsrc(a.exch + (long long)b * XcdExch::WORDS, 0, (int)(XcdExch::WORDS * 8), 0x00020000);
}
// the same view with the descriptors made on demand (many-streams kernel: twelve descriptors held in scalar registers for the whole
// launch spilled); slots of streams the XCD does not have alias stream 0
struct XLazyRs { const XcdLaunch* a; int xcc, ns; __device__ __forceinline__ rsrc_t operator[](int k) const { return exch_rsrc(*a, xcc + 8 * (k < ns ? k : 0)); } };
struct XLazyB { int xcc, ns; __device__ __forceinline__ int operator[](int k) const { return xcc + 8 * (k < ns ? k : 0); } };
template <int NS> struct XStreamsLazy { XLazyRs rs; XLazyB b; };
constexpr int kSkipLdsWords = 32 * 64;       // 8-byte LDS words per stream in a skip workgroup
constexpr int kConvLdsFloats = 16 * 64 + 64; // LDS floats per stream in a conv1 workgroup

// =====================================================================================================================
//  CHAIN workgroup: model.py:41-46 causal layer (wave 0), model.py:66-101 residual layers (relay over the waves),
//  mixture.py:84-114 sampler (wave 7)
// =====================================================================================================================
// ONEHOT (scalar_input False, the mu-law-256 model of generate.py:219-231): the head wave feeds the causal layer with two kernel ROWS
// (model.py:41-46 over one-hot input: every AC-1 chunk holds at most one non-zero term, so the k = 2 conv is W0[q(t-1)] + W1[q(t)];
// only the W1 row waits for the sample) and draws the next class with twv_categorical.hpp from the 256 logits the conv1 workgroups publish.
// HW: 1 = this wave is the head (wave 7 of the first chain workgroup), 0 = it is not, -1 = decided at run time.  Known at compile time
// the instantiations of waves 0..6 carry none of the head's code (causal layer, sampler) and may defer their granule stores (DEFER).
template <int INSTR, bool ALL, bool FORCED, bool SEG1, bool TWOSEG, int NC, bool ONEHOT = false, int HW = -1>
__device__ __forceinline__ void chain_role(const XArgs& xa, int b, rsrc_t rs)
{
    const XcdLaunch& a = xa.p;
    const Layout& L = a.lay;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int NL = L.NL, T = a.T;
    const bool use_bias = L.use_bias != 0, has_gc = L.G > 0, has_lc = L.L > 0;
    constexpr bool forced = FORCED;                           // teacher-forced steps (twv_wavenet_prime): its own instantiation, the sampling loop carries none of it
    // first chain workgroup: waves 0..5 hold four layers each, waves 6 and 7 three: wave 7 also runs the sampler and the causal
    // layer (it HAS the new sample).  A model with more than 30 layers (hparams.py's default has 50) continues in a second chain
    // workgroup (SEG1: four layers per wave from layer 30 on), one L2 hop away.
    const int l0 = SEG1 ? kXcdSeg0Layers + 4 * w : (w < 6 ? 4 * w : 24 + 3 * (w - 6));
    const int lw0 = SEG1 ? 4 * w : l0;                         // the wave's first layer, counted inside this workgroup (LDS copies)
    const int cap = SEG1 ? 4 : (w < 6 ? 4 : 3);
    int nl = NL - l0;
    nl = nl < 0 ? 0 : (nl > cap ? cap : nl);
    // NC: the wave's layer count as a compile-time constant (4 or 3; -1: the run-time value) -- the step loop then carries no dispatch
    // on the count (three compare-and-branch pairs on the wave-to-wave hand-off path)
    const int nlc = NC >= 0 ? NC : nl;
    const bool next_has = (l0 + nl < NL);                      // a later wave continues the stack
    const bool head = HW >= 0 ? (HW == 1) : (!SEG1 && (w == 7));      // sampler + causal layer
    const bool to_seg1 = TWOSEG && !SEG1 && w == 7 && next_has;     // the stack goes on in the second chain workgroup
    constexpr bool two_seg = TWOSEG;
    if (nl == 0 && !head) return;
    const int oc = dpp_conv_out(lane), od = dpp_dense_out(lane);
    const ActCoef coef = act_coef(lane >= 32);
    float* stb = a.state + (long long)b * L.state_stride;
    Poll pl{rs, a.status, 0, false};
    constexpr int O_WD = 2048;                                 // LDS floats: dense kernels [layer][4][64 lanes][4]

    // ---- the wave's layers: tap-1 conv kernel register-resident for the whole launch, dense kernel LDS-resident
    ChainRegs W[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < nl) {
            const f32x4* src = reinterpret_cast<const f32x4*>(a.P + L.off_xl + (long long)(l0 + i) * kXcdXlFloats) + lane;
#pragma unroll
            for (int q = 0; q < 8; ++q) { const f32x4 v = src[q * 64]; W[i].wc[4 * q] = v.x; W[i].wc[4 * q + 1] = v.y; W[i].wc[4 * q + 2] = v.z; W[i].wc[4 * q + 3] = v.w; }
#pragma unroll
            for (int q = 0; q < 4; ++q) LDS4((O_WD >> 2) + ((lw0 + i) * 4 + q) * 64 + lane) = src[(8 + q) * 64];
            const f32x4 v = src[12 * 64];
            W[i].bd = dense_bias_init(lane, use_bias ? v.y : 0.0f);
        }
    }
    // wave 7: causal kernel (model.py:41-46), every lane the 32 taps of ITS residual channel (X layout), in the registers of the
    // fourth layer slot (wave 7 holds three layers); causal queue (model.py:52) as two row-broadcast registers:
    // ha: every row hist[n], hb: every row hist[16+n]  (hist[31] = newest input)
    float ha = 0.0f, hb = 0.0f, first_in = 0.0f, b2v = 0.0f, s_lnl = 0.0f, s_tq = 0.0f, samp = 0.0f;
    float cp[4] = {0.0f, 0.0f, 0.0f, 0.0f};                   // the causal chunk without its newest term
    // one-hot model: previous input class (the tap-0 row is fetched a step ahead), the class just drawn, the step's uniform draw
    int q_prev = 0, q_valid = 0, samp_q = 0, first_q = 0;
    float w0row = 0.0f;
    double u_next = 0.0;
    const int chx = dpp_dense_out(lane);                      // residual channel of this lane in the X layout
    const bool sampler = head && !forced;
    const bool is15 = (lane & 15) == 15;
    // model.py:122 queue shift (the slot of the newest sample stays open) + the 31 terms that do not need it
    auto causal_prepare = [&]() {
        const float t1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ha), 0x101, 0xf, 0xf, true));   // row_shl:1
        const float b0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(hb), 0x150, 0xf, 0xf, true));   // row_newbcast:0
        ha = is15 ? b0 : t1;
        hb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(hb), 0x101, 0xf, 0xf, true));
        causal_partial_dpp(W[3].wc, ha, hb, cp);
    };
    auto noise = [&](int t) {
        // mixture.py:103 -log(-log u) per mixture lane; mixture.py:110-111 log u - log(1 - u) of the last draw
        const float* up = a.uniforms + ((long long)b * T + t) * (L.nr_mix + 1);
        const float u = lane <= L.nr_mix ? up[lane] : 0.5f;
        s_lnl = log_e(-log_e(u));
        const float uu = __shfl(u, L.nr_mix);
        s_tq = log_e(uu) - log_e(1.0f - uu);
    };
    if (head) {
        if constexpr (ONEHOT) {
            const int* meta = reinterpret_cast<const int*>(stb + L.st_meta);
            q_valid = __builtin_amdgcn_readfirstlane(meta[M_PREV_VALID]);      // model.py:52 causal queue: zeros after queue_initializer
            q_prev = __builtin_amdgcn_readfirstlane(meta[M_QPREV]) & (L.Q - 1);
            w0row = a.P[L.off_causal + (long long)q_prev * 32 + chx];
            if (!forced) first_q = reinterpret_cast<const int*>(a.first_input)[b];
        } else {
            const f32x4* src = reinterpret_cast<const f32x4*>(a.P + L.off_xc) + lane;
#pragma unroll
            for (int q = 0; q < 8; ++q) { const f32x4 v = src[q * 64]; W[3].wc[4 * q] = v.x; W[3].wc[4 * q + 1] = v.y; W[3].wc[4 * q + 2] = v.z; W[3].wc[4 * q + 3] = v.w; }
            ha = stb[L.st_hist + (lane & 15)];
            hb = stb[L.st_hist + 16 + (lane & 15)];
            if (!forced) first_in = reinterpret_cast<const float*>(a.first_input)[b];
            if (sampler && use_bias && lane < L.O) b2v = a.P[L.off_b2 + lane];
            causal_prepare();
        }
    }
    __builtin_amdgcn_s_waitcnt(0);        // every register image and LDS copy has landed before the relay starts
    // The step loop starts at a fixed offset inside a 64-byte fetch window: the same instructions with the same registers ran at
    // 10.42 or 10.60 us/step depending on where unrelated code had pushed them (no instruction-cache misses either way; aligning
    // every loop head gives the slow figure).  TWV_CHAIN_PAD nops after the alignment: scanned, see DESIGN.md section 4.
    asm volatile(".p2align 6\n\t.rept " TWV_STR(TWV_CHAIN_PAD) "\n\ts_nop 0\n\t.endr" ::: "memory");

    float X = 0.0f;
    unsigned long long t_in = 0, in_period = 0, now_in = 0;   // when this wave's input arrived in the previous step, the step period
    for (int t = 0; t < T && !pl.dead; ++t) {
        asm volatile(".p2align 6");                        // (XALIGN, see chain_many_role)
        const unsigned tag = (unsigned)t + 1u;
        // ---- wave 7, head of the step: new input sample -> causal layer -> wave 0.  On the sample-to-sample path: one fma, three adds.
        if (head) {
            if (forced && t > 0) {
                pl.it = 0;
                for (;;) {
                    const unsigned long long q = two_seg ? xb_load(rs, (int)XcdExch::DONE, lane) : LDSU64(8 * 64 + lane);
                    if (__all(g_tag(q) == (unsigned)t)) break;             // the tag of step t-1
                    if (!poll_tick(pl, 35)) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                if (pl.dead) break;
            }
            XSTAMP(true, 0);
            if constexpr (ONEHOT) {
                const int q_raw = forced ? reinterpret_cast<const int*>(a.forced)[(long long)b * T + t] : (t == 0 ? first_q : samp_q);
                const int q_in = __builtin_amdgcn_readfirstlane(q_raw) & (L.Q - 1);
                const float w1row = a.P[L.off_causal + ((long long)L.Q + q_in) * 32 + chx];       // on the sample path: one row load, one add
                const float x0 = q_valid ? w0row + w1row : w1row;       // model.py:41-46; the queue's older slot is empty after a reset
                LDSU64(0 * 64 + lane) = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(x0);
                XSTAMP(true, 1);
                if (lane == 0) {
                    if (sampler && t > 0) reinterpret_cast<int*>(a.out)[(long long)b * T + t - 1] = samp_q;
                    xb_store(rs, (int)XcdExch::CTRL, 0, tag, 0.0f);    // step t has started (the lc workgroups throttle on it)
                }
                q_prev = q_in; q_valid = 1;                             // model.py:122 queue shift
                w0row = a.P[L.off_causal + (long long)q_in * 32 + chx]; // the next step's tap-0 row, a step ahead
                if (sampler) u_next = reinterpret_cast<const double*>(a.uniforms)[(long long)b * T + t];      // generate.py:231's draw
            } else {
            const float s_in = forced ? reinterpret_cast<const float*>(a.forced)[(long long)b * T + t] : (t == 0 ? first_in : samp);
            const float c3 = fma_(W[3].wc[31], s_in, cp[3]);            // k = 31, the last term of chain 3
            const float x0 = (cp[0] + cp[1]) + (cp[2] + c3);            // model.py:41-46: one AC-1 chunk, no bias; X layout
            LDSU64(0 * 64 + lane) = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(x0);
            XSTAMP(true, 1);
            // off the path: results out, progress, the queue takes the sample, next step's partial chunk and noise terms
            if (lane == 0) {
                if (sampler && t > 0) a.out[(long long)b * T + t - 1] = samp;
                xb_store(rs, (int)XcdExch::CTRL, 0, tag, 0.0f);        // step t has started (the lc workgroups throttle on it)
            }
            hb = is15 ? s_in : hb;                                      // (ha, hb) = the queue after step t
            if (t + 1 < T) causal_prepare();
            if (sampler) noise(t);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        XMARK(SEG1 ? 30 : ROLE_CHAIN, 1);
        // ---- (A) this step's addends of the wave's layers: ((tap-0 chunk + bias) + gc) + lc (service workgroup; long since published)
        float pre[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (nlc > 0) {
            pl.it = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < nlc) {
                        const unsigned long long qp = xb_load(rs, (int)XcdExch::PG + (l0 + i) * 64, oc);
                        ok = ok && g_tag(qp) == tag;
                        pre[i] = g_val(qp);
                    }
                }
                if (__all(ok)) break;
                if (!poll_tick(pl, 31)) break;
                __builtin_amdgcn_s_sleep(4);
            }
            // (a watchdog abort ends the loop at its head: an exit here would make the compiler thread state flags through the hand-off path)
            XSTAMP(!SEG1, 26 + w);
            XMARK(SEG1 ? 30 : ROLE_CHAIN, 2);
            // ---- (B) the wave's input: the previous wave's residual vector (wave 0: the causal layer's output).  The wave sleeps
            // through most of the step and polls only when its turn is near: a spinning wave takes issue slots from the wave that
            // shares its SIMD (waves w and w+4), and that one may be the wave carrying the chain right now.
            if (in_period) nap_until(t_in + in_period - (in_period >> 4));
            unsigned long long q;
            pl.it = 0;
            if (SEG1 && w == 0) {
                for (;;) {                                               // from wave 7 of the first chain workgroup: one L2 hop
                    q = xb_load(rs, (int)XcdExch::SEG, lane);
                    if (__all(g_tag(q) == tag)) break;
                    if (!poll_tick(pl, 36)) break;
                }
            } else {
                for (;;) {
                    q = LDSU64(w * 64 + lane);
                    if (__all(g_tag(q) == tag)) break;
                    if (!poll_tick(pl, 33)) break;
                }
            }
            // (a watchdog abort ends the loop at its head: an exit here would make the compiler thread state flags through the hand-off path)
            __builtin_amdgcn_s_setprio(3);
            X = g_val(q);
            now_in = __builtin_amdgcn_s_memtime();             // only READ here: the arithmetic on it waits until the layers are out (the
                                                               // clock's bookkeeping on the hand-off path cost 0.2 us per step: 10.40 -> 10.18)
            XSTAMP(true, (SEG1 ? 40 : 2) + w);
            XMARK(SEG1 ? 30 : ROLE_CHAIN, 3);
        }
        // ---- the wave's layers (the loop is compiled once per layer count: a run-time count costs a branch pair per layer)
        // DEFER (round 5): the granule store of a layer costs the wave ~40 issue cycles (tuple assembly, the 1 KB of store data, the
        // hazard wait: profiles/r05_chain_contract_ubench.txt, shape P against N) and nobody needs z / the layer input of an EARLY wave
        // at once -- the skip workgroups' sum only completes with the LAST layer, the service workgroup works a step ahead -- so the
        // waves that are not the head (HW == 0: waves 0..6) keep {z, input} in registers and store their granules behind the hand-off
        // to the next wave, off the sample-to-sample path.  The head wave (the stack's last layers) stores at once.
        constexpr bool DEFER = ALL && !FORCED && !SEG1 && NC > 0 && HW == 0;
        float zs[4] = {0.0f, 0.0f, 0.0f, 0.0f}, xs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        auto run_layers = [&](auto nc) {
            constexpr int N = decltype(nc)::value;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                float wd[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = LDS4((O_WD >> 2) + ((lw0 + i) * 4 + q) * 64 + lane);
                    wd[4 * q] = v.x; wd[4 * q + 1] = v.y; wd[4 * q + 2] = v.z; wd[4 * q + 3] = v.w;
                }
                const float z = layer_front_dpp(W[i].wc, coef, X, pre[i]);
                // one 16-byte store, two self-tagged halves: {z, tag} -> skip workgroups, {layer input, tag} -> service workgroup
                // (model.py:145: the queue takes the layer INPUT)
                if (DEFER) { zs[i] = z; xs[i] = X; }
                else xb_store2(rs, (int)XcdExch::ZX + (l0 + i) * 128, lane, tag, z, X);
                layer_back_dpp(wd, W[i].bd, z, X);
                if ((INSTR & 2) && a.dbg != nullptr && t < a.dbg_steps) {
                    float* dp = a.dbg + ((long long)b * a.dbg_steps + t) * ((long long)NL * 64 + L.Opad) + (long long)(l0 + i) * 64;
                    if (lane < 32) dp[dpp_z_index(lane)] = z;
                    if ((lane & 16) == 0) dp[32 + od] = X;
                }
            }
        };
        if constexpr (NC > 0) run_layers(std::integral_constant<int, NC>{});
        else {
            if (nl == 4) run_layers(std::integral_constant<int, 4>{});
            else if (nl == 3) run_layers(std::integral_constant<int, 3>{});
            else if (nl == 2) run_layers(std::integral_constant<int, 2>{});
            else if (nl == 1) run_layers(std::integral_constant<int, 1>{});
        }
        if (to_seg1) xb_store(rs, (int)XcdExch::SEG, lane, tag, X);
        else if (nlc > 0) LDSU64((next_has ? w + 1 : 9) * 64 + lane) = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(X);   // nobody reads box 9: no branch on the hand-off
        XSTAMP(DEFER && nl > 0, (SEG1 ? 48 : 10) + w);       // (instrumented build: "layers done" = the hand-off, in front of the deferred stores)
        if (DEFER) {
#pragma unroll
            for (int i = 0; i < (NC > 0 ? NC : 0); ++i) xb_store2(rs, (int)XcdExch::ZX + (l0 + i) * 128, lane, tag, zs[i], xs[i]);
        }
        // teacher-forced steps (twv_wavenet_prime): nothing makes the head wait for the stack (there is no sample to wait for), so
        // the wave that runs the last layer reports the end of the step in box 8 and the head starts the next one after that
        if (forced && nl > 0 && !next_has) {
            if (SEG1) xb_store(rs, (int)XcdExch::DONE, lane, tag, 0.0f);
            else LDSU64(8 * 64 + lane) = (unsigned long long)tag << 32;
        }
        if (nlc > 0) {                                         // the wave's clock for its next nap (off the hand-off path)
            const unsigned long long d = now_in - t_in;
            in_period = (t_in != 0 && d < (1ull << 18)) ? d : 0;
            t_in = now_in;
        }
        if (!sampler) __builtin_amdgcn_s_setprio(0);
        XSTAMP(!DEFER && nl > 0, (SEG1 ? 48 : 10) + w);
        XMARK(SEG1 ? 30 : ROLE_CHAIN, 4);
        // ---- sampler: conv1d_2's [16 chunks][32 lanes] partial table from the conv1 workgroups -> mixture.py:84-114
        if constexpr (ONEHOT) {
          if (sampler) {
            __builtin_amdgcn_s_setprio(3);
            // the 256 logits (conv1d_2 summed in chunk order + bias by the conv1 workgroups): class lane + 64 k in granule 4 lane + k,
            // two 16-byte loads per lane
            u32x4s d0, d1;
            pl.it = 0;
            for (;;) {
                d0 = xb_load2(rs, (int)XcdExch::QL, lane * 2);
                d1 = xb_load2(rs, (int)XcdExch::QL, lane * 2 + 1);
                if (__all(d0.y == tag && d0.w == tag && d1.y == tag && d1.w == tag)) break;
                if (!poll_tick(pl, 34)) break;
            }
            XSTAMP(true, 18);
            XMARK(SEG1 ? 30 : ROLE_CHAIN, 5);
            const float y[4] = {__uint_as_float(d0.x), __uint_as_float(d0.z), __uint_as_float(d1.x), __uint_as_float(d1.z)};
            if ((INSTR & 2) && a.dbg != nullptr && t < a.dbg_steps) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    a.dbg[((long long)b * a.dbg_steps + t) * ((long long)NL * 64 + L.Opad) + (long long)NL * 64 + lane + 64 * k] = y[k];
            }
            // model.py:243 + generate.py:219-231 (AC-5, twv_categorical.hpp): the drawn class is the next step's input
            bool bad_p = false;
            samp_q = categorical_sample<4>(y, L.Q, lane, a.temperature, u_next, nullptr,
                                           ((INSTR & 1) && a.prof != nullptr && b == a.prof_stream && t < a.prof_steps) ? a.prof + (long long)t * 64 + 38 : nullptr, &bad_p);
            if (bad_p && lane == 0) atomicMax(a.status, 31);       // NaN probabilities: np.random.choice would raise (generate.py:231)
            XSTAMP(true, 19);
          }
        } else
        if (sampler) {
            __builtin_amdgcn_s_setprio(3);
            // the [8 chunk pairs][32 outputs][2] partial table, each granule once, FOUR 16-byte loads per round (every load in a polling
            // round adds to its round trip: 16 8-byte loads per lane were measured at +0.2 us on the hop over 8): lanes 0-31 chunks 0-7
            // of output (lane & 31), lanes 32-63 chunks 8-15
            const int half = lane >> 5;
            unsigned long long q[8];
            pl.it = 0;
            for (;;) {
                bool good = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u32x4s d = xb_load2(rs, (int)XcdExch::PT, (half * 4 + k) * 32 + (lane & 31));
                    q[2 * k] = ((unsigned long long)d.y << 32) | d.x;
                    q[2 * k + 1] = ((unsigned long long)d.w << 32) | d.z;
                    good = good && d.y == tag && d.w == tag;
                }
                if (__all(good)) break;
                if (!poll_tick(pl, 34)) break;
            }
            // (a watchdog abort ends the loop at its head: an exit here would make the compiler thread state flags through the hand-off path)
            XSTAMP(true, 18);
            XMARK(SEG1 ? 30 : ROLE_CHAIN, 5);
            float y = g_val(q[0]);                                     // chunk partials added in chunk order (AC-1)
#pragma unroll
            for (int k = 1; k < 8; ++k) y = y + g_val(q[k]);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)q[k], (unsigned)q[k], false, false);
                y = y + __uint_as_float(sw[1]);                        // chunk 8 + k of the same output (upper half-wave's granule)
            }
            if (use_bias && lane < L.O) y = y + b2v;
            if ((INSTR & 2) && a.dbg != nullptr && t < a.dbg_steps)
                a.dbg[((long long)b * a.dbg_steps + t) * ((long long)NL * 64 + L.Opad) + (long long)NL * 64 + lane] = y;
            const int nr = L.nr_mix;
            // mixture.py:107 exp(max(log_scale, log 1e-14)) on EVERY lane (lane 2nr+i holds log_scale_i), next to the argmax chain
            const float lsmin = (float)-32.23619130191664;
            const float e_all = exp_e(y > lsmin ? y : lsmin);
            // mixture.py:103 argmax_i(logit_i - log(-log u_i)), first maximum: running maximum over lanes 0..15 with four
            // v_max_f32 row_shr steps, then the lowest lane that equals it
            const float ninf = __uint_as_float(0xff800000u);
            const float gmb = (lane < nr) ? y - s_lnl : ninf;
            float mx = gmb;
            mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ninf), __float_as_int(mx), 0x111, 0xf, 0xf, false)));
            mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ninf), __float_as_int(mx), 0x112, 0xf, 0xf, false)));
            mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ninf), __float_as_int(mx), 0x114, 0xf, 0xf, false)));
            mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ninf), __float_as_int(mx), 0x118, 0xf, 0xf, false)));
            const float best = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mx), 15));
            const unsigned long long hit = __ballot(lane < nr && gmb == best);
            const int k = hit ? (int)__ffsll((long long)hit) - 1 : 0;
            const float mean = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), nr + k));        // mixture.py:105
            const float e = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e_all), 2 * nr + k));    // mixture.py:107
            const float prod = e * s_tq;                               // mixture.py:110-111
            float xs = mean + prod;
            xs = xs > -1.0f ? xs : -1.0f;                              // mixture.py:113
            xs = xs < 1.0f ? xs : 1.0f;
            samp = xs;                                                 // uniform: the next step's input, straight into the causal layer
            XSTAMP(true, 19);
        }
    }

    // ---- persist (model.py:49-64 causal queue): canonical order, element k = k-th oldest input
    if (head) {
        if constexpr (ONEHOT) {
            if (lane == 0) {
                if (sampler && !pl.dead && T > 0) reinterpret_cast<int*>(a.out)[(long long)b * T + T - 1] = samp_q;
                int* meta = reinterpret_cast<int*>(stb + L.st_meta);
                meta[M_TABS] = meta[M_TABS] + T;
                meta[M_PREV_VALID] = q_valid;
                meta[M_QPREV] = q_prev;
            }
        } else {
        if (sampler && !pl.dead && T > 0 && lane == 0) a.out[(long long)b * T + T - 1] = samp;
        if (lane < 16) stb[L.st_hist + lane] = ha;
        else if (lane < 32) stb[L.st_hist + lane] = hb;
        if (lane == 0) {
            int* meta = reinterpret_cast<int*>(stb + L.st_meta);
            meta[M_TABS] = meta[M_TABS] + T;
        }
        }
    }
}

// =====================================================================================================================
//  SERVICE workgroup: the delay lines (model.py:49-64, 116-126, 144-146) and the tap-0 chunk of conv_filter|conv_gate
//  (model.py:68-69: the tap that reads x[t-d]), one step ahead of the chain; forwards the lc projections.
//  Wave s owns layers s, s+8, s+16, s+24.
// =====================================================================================================================
// a weight tile kept in LDS (layers beyond the register slots of a helper wave): same [8 float4][64 lanes] image; the dot takes it
// in two halves of 16 registers (the same two blocks of 16 v_fmac_f32_dpp as dot32_dpp, twv_dpp.hpp)
#define TWV_FMAC16_DPP(c0, c1, c2, c3, x, w)                                                          \
    asm volatile(                                                                                      \
        "s_nop 1\n" TWV_ALIGN8                                                                         \
        "v_fmac_f32_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"                        \
        "v_fmac_f32_dpp %1, %4, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"                        \
        "v_fmac_f32_dpp %2, %4, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"                        \
        "v_fmac_f32_dpp %3, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"                        \
        "v_fmac_f32_dpp %0, %4, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"                        \
        "v_fmac_f32_dpp %1, %4, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"                       \
        "v_fmac_f32_dpp %2, %4, %11 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"                       \
        "v_fmac_f32_dpp %3, %4, %12 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"                       \
        "v_fmac_f32_dpp %0, %4, %13 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"                       \
        "v_fmac_f32_dpp %1, %4, %14 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"                       \
        "v_fmac_f32_dpp %2, %4, %15 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"                      \
        "v_fmac_f32_dpp %3, %4, %16 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"                      \
        "v_fmac_f32_dpp %0, %4, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"                      \
        "v_fmac_f32_dpp %1, %4, %18 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"                      \
        "v_fmac_f32_dpp %2, %4, %19 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"                      \
        "v_fmac_f32_dpp %3, %4, %20 row_newbcast:15 row_mask:0xf bank_mask:0xf"                        \
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)                                                       \
        : "v"(x), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), \
          "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]))
__device__ __forceinline__ float dot32_dpp_lds(int off4, int lane, float xa, float xb)
{
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float w[16];
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const f32x4 q = LDS4(off4 + (4 * h + kq) * 64 + lane);
            w[4 * kq + 0] = q.x; w[4 * kq + 1] = q.y; w[4 * kq + 2] = q.z; w[4 * kq + 3] = q.w;
        }
        if (h == 0) { TWV_FMAC16_DPP(c0, c1, c2, c3, xa, w); }
        else { TWV_FMAC16_DPP(c0, c1, c2, c3, xb, w); }
    }
    return (c0 + c1) + (c2 + c3);
}
__device__ __forceinline__ void copy_tile_to_lds(int off4, const float* base, int lane)
{
    const f32x4* p = reinterpret_cast<const f32x4*>(base) + lane;
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) LDS4(off4 + kq * 64 + lane) = p[kq * 64];
}

__device__ __forceinline__ unsigned ring_slot(unsigned pos0, unsigned t, unsigned d)
{
    const unsigned v = pos0 + t;
    return (d & (d - 1)) == 0 ? (v & (d - 1)) : v % d;
}
template <int INSTR, bool BIG>
__device__ __forceinline__ void service_role(const XArgs& xa, int b, rsrc_t rs)
{
    const XcdLaunch& a = xa.p;
    const Layout& L = a.lay;
    const int s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int NL = L.NL, T = a.T;
    const bool has_lc = L.L > 0, use_bias = L.use_bias != 0, has_gc = L.G > 0;
    float* stb = a.state + (long long)b * L.state_stride;
    float* ring = stb + L.st_ring;
    const int* pmeta = reinterpret_cast<const int*>(a.P + L.off_meta);
    int* ringpos = reinterpret_cast<int*>(stb + L.st_ringpos);
    Poll pl{rs, a.status, 0, false};
    // BIG (more than 32 layers): four tiles in registers (layers NLDS + s + 8i), the first NLDS = NL - 32 layers' tiles in LDS (this
    // workgroup has no other use for it: 18 tiles = 144 KiB at 50 layers)
    constexpr int kReg = 4, kLds = BIG ? 3 : 0, kSlots = kReg + kLds;
    const int NLDS = BIG ? (NL > 8 * kReg ? NL - 8 * kReg : 0) : 0;
    int nown = 0, nlds = 0;
#pragma unroll
    for (int i = 0; i < kReg; ++i) if (NLDS + s + 8 * i < NL) nown = i + 1;
#pragma unroll
    for (int j = 0; j < kLds; ++j) if (s + 8 * j < NLDS) nlds = j + 1;
    if (nown == 0 && nlds == 0) return;
    Tile t0[kReg];
    unsigned dil[kSlots], roff[kSlots], pos0[kSlots];
    int lay_of[kSlots];
    float bfg[kSlots], gcv[kSlots];          // conv bias (model.py:68-69) and the hoisted gc projection (model.py:71-73) of this lane's conv output
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
        dil[i] = 1; roff[i] = 0; pos0[i] = 0; bfg[i] = 0.0f; gcv[i] = 0.0f;
        const bool is_lds = i < kLds;
        const int l = is_lds ? s + 8 * i : NLDS + s + 8 * (i - kLds);
        const bool have = is_lds ? i < nlds : (i - kLds) < nown;
        lay_of[i] = have ? l : -1;
        if (have) {
            const float* src = a.P + L.off_layer0 + (long long)l * L.layer_stride + LayerOff::T0;
            if (is_lds) copy_tile_to_lds(l * (kTile / 4), src, lane);     // own layers only: written and read by this wave alone
            else load_tile(t0[i - kLds < 0 ? 0 : i - kLds], src, lane);
            dil[i] = (unsigned)pmeta[l]; roff[i] = (unsigned)pmeta[64 + l]; pos0[i] = (unsigned)ringpos[l];
            if (use_bias) bfg[i] = a.P[L.off_layer0 + (long long)l * L.layer_stride + LayerOff::BFG + lane];
            if (has_gc) gcv[i] = a.cond[XH_WORDS + ((long long)b * NL + l) * 64 + lane];
        }
    }
    // AC-1b: what the chain starts its tap-1 chunk from -- the reference's statement order ((conv + bias) + gc) + lc (model.py:68-83)
    // with conv = chunk(tap 0) + chunk(tap 1): everything but the tap-1 chunk, summed here, a step ahead of the chain
    auto addend = [&](int i, float pre, float lcv) __attribute__((always_inline)) -> float {
        float v = pre;
        if (use_bias) v = v + bfg[i];
        if (has_gc) v = v + gcv[i];
        if (has_lc) v = v + lcv;
        return v;
    };
    const int n16 = lane & 15;
    // operand of the tap-0 chunk of step t: x[t-d] = the slot the delay line overwrites at step t
    auto tap0 = [&](int i, unsigned t, float& xa_, float& xb_) __attribute__((always_inline)) {
        const unsigned slot = ring_slot(pos0[i], t, dil[i]);
        const unsigned long long* p = reinterpret_cast<const unsigned long long*>(ring + roff[i] + slot * 32);
        // 8-byte sc1 loads of float pairs: lanes n and n+1 of a pair read the same word
        const unsigned long long qa = __hip_atomic_load((gu64*)(p + (n16 >> 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long qb = __hip_atomic_load((gu64*)(p + 8 + (n16 >> 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        xa_ = __uint_as_float((n16 & 1) ? (unsigned)(qa >> 32) : (unsigned)qa);
        xb_ = __uint_as_float((n16 & 1) ? (unsigned)(qb >> 32) : (unsigned)qb);
    };
    // the tile of slot i: a register tile, or (BIG, early layers) fetched from LDS
    auto dot_slot = [&](int i, float oa, float ob) __attribute__((always_inline)) -> float {
        if (i < kLds) return dot32_dpp_lds(lay_of[i] * (kTile / 4), lane, oa, ob);
        return dot32_dpp(t0[i - kLds < 0 ? 0 : i - kLds].w, oa, ob);
    };
    // ---- step 0: from the persisted state
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
        if (lay_of[i] >= 0) {
            const int l = lay_of[i];
            float xa_, xb_;
            tap0(i, 0u, xa_, xb_);
            const float pre = dot_slot(i, xa_, xb_);
            const float lcv = has_lc ? stb[L.st_lcprev + l * 64 + lane] : 0.0f;     // frame pushed by the previous call (zeros after reset)
            xb_store(rs, (int)XcdExch::PG + l * 64, lane, 1u, addend(i, pre, lcv));
        }
    }
    for (int t = 0; t < T && !pl.dead; ++t) {
        asm volatile(".p2align 6");                        // (XALIGN, see chain_many_role)
        const unsigned tag = (unsigned)t + 1u;
#pragma unroll
        for (int i = 0; i < kSlots; ++i) {
            if (lay_of[i] >= 0 && !pl.dead) {
                const int l = lay_of[i];
                const bool more = t + 1 < T;
                // next step's operand when it is already in the delay line (d >= 2): requested before the wait
                float oa = 0.0f, ob = 0.0f;
                if (more && dil[i] > 1) tap0(i, (unsigned)t + 1u, oa, ob);
                XMARK(ROLE_SERVICE, 10 + i);
                // the layer input x_l[t] from the chain
                unsigned long long qa, qb;
                pl.it = 0;
                for (;;) {
                    qa = xb_load(rs, (int)XcdExch::ZX + l * 128 + 1, n16 * 2);
                    qb = xb_load(rs, (int)XcdExch::ZX + l * 128 + 65, n16 * 2);
                    if (__all(g_tag(qa) == tag && g_tag(qb) == tag)) break;
                    if (!poll_tick(pl, 41)) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                if (pl.dead) break;
                const float xa_ = g_val(qa), xb_ = g_val(qb);
                // model.py:145 dilation queue <- layer input
                const unsigned slot = ring_slot(pos0[i], (unsigned)t, dil[i]);
                if (lane < 32) ring[roff[i] + slot * 32 + lane] = lane < 16 ? xa_ : xb_;
                if (more) {
                    if (dil[i] == 1) { oa = xa_; ob = xb_; }
                    const float pre = dot_slot(i, oa, ob);
                    float lcv = 0.0f;
                    XMARK(ROLE_SERVICE, 20 + i);
                    if (has_lc) {
                        // lc frame used at step t+1 = frame pushed at step t (model.py:79-80: slice from the FRONT of the queue)
                        const int lw = (int)XcdExch::LCR + (((t + 1) % kXcdLcRing) * kXcdLs + l) * 64;
                        unsigned long long ql;
                        pl.it = 0;
                        for (;;) {
                            ql = xb_load(rs, lw, lane);
                            if (__all(g_tag(ql) == tag + 1u)) break;
                            if (!poll_tick(pl, 42)) break;
                            __builtin_amdgcn_s_sleep(8);
                        }
                        if (pl.dead) break;
                        lcv = g_val(ql);
                    }
                    xb_store(rs, (int)XcdExch::PG + l * 64, lane, tag + 1u, addend(i, pre, lcv));
                }
            }
        }
    }
    if (lane == 0 && !pl.dead) {
#pragma unroll
        for (int i = 0; i < kSlots; ++i)
            if (lay_of[i] >= 0) ringpos[lay_of[i]] = (int)((pos0[i] + (unsigned)T) % dil[i]);
    }
}

// =====================================================================================================================
//  SKIP workgroup g: model.py:94-96 skip 1x1 of every layer for output block g, model.py:154 sum over the layers in layer
//  order, model.py:157 relu.  Wave v owns layers v, v+8, ...; the wave that owns the last layer adds the values up as they appear.
// =====================================================================================================================
// BIG (more than 32 layers; hparams.py's default has 50): a wave holds FIVE tiles in registers -- the layers NLDS + v + 8i, the late
// ones, whose values are needed soonest after they appear -- and the tiles of the first NLDS = NL - 40 layers (v, v + 8) sit in LDS
// and pass through a sixth register tile when their turn comes.
// KF (round 6): the LAST KF layers' skip 1x1 is not computed here but in the conv1 workgroups (conv1_role<..., KF>), which receive z of
// those layers straight from the chain: the skip -> conv1 hop then only carries the running sum of the layers in front of them -- raw,
// the relu follows the last add over there -- and leaves the path behind the last layer.
template <int INSTR, int NS, bool BIG, int KF = 0>
__device__ __forceinline__ void skip_role(const XArgs& xa, const XStreams<NS>& sx, int g)
{
    const XcdLaunch& a = xa.p;
    const Layout& L = a.lay;
    const int v = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int NL = L.NL - KF, T = a.T;                       // (the layers this role serves)
    const bool use_bias = L.use_bias != 0;
    Poll pl{sx.rs[0], a.status, 0, false};
    constexpr int kReg = BIG ? 5 : 4, kLds = BIG ? 2 : 0, kSlots = kReg + kLds;
    const int NLDS = BIG ? (NL > 8 * kReg ? NL - 8 * kReg : 0) : 0;      // layers whose tiles live in LDS
    const int SW = BIG ? NL * 64 : kSkipLdsWords;                        // 8-byte value slots per stream
    const int o_tile4 = (NS * SW * 2 + 3) >> 2;                          // LDS tiles behind the value slots (float4 units)
    int nown = 0, nlds = 0;
#pragma unroll
    for (int i = 0; i < kReg; ++i) if (NLDS + v + 8 * i < NL) nown = i + 1;
#pragma unroll
    for (int j = 0; j < kLds; ++j) if (v + 8 * j < NLDS) nlds = j + 1;
    if (nown == 0 && nlds == 0) return;
    Tile ws[kReg];
    float bs[kSlots];
#pragma unroll
    for (int i = 0; i < kSlots; ++i) bs[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < kReg; ++i) {
        if (i < nown) {
            const long long lb = L.off_layer0 + (long long)(NLDS + v + 8 * i) * L.layer_stride;
            load_tile(ws[i], a.P + lb + LayerOff::SK + (long long)g * kTile, lane);
            if (use_bias) bs[kLds + i] = a.P[lb + LayerOff::SK + (long long)L.NSJ * kTile + g * 64 + lane];
        }
    }
#pragma unroll
    for (int j = 0; j < kLds; ++j) {
        if (j < nlds) {                                                   // own layers only: written and read by this wave alone
            const long long lb = L.off_layer0 + (long long)(v + 8 * j) * L.layer_stride;
            copy_tile_to_lds(o_tile4 + (v + 8 * j) * (kTile / 4), a.P + lb + LayerOff::SK + (long long)g * kTile, lane);
            if (use_bias) bs[j] = a.P[lb + LayerOff::SK + (long long)L.NSJ * kTile + g * 64 + lane];
        }
    }
    // the wave that owns the last layer adds a stream's values up.  With three or four streams per XCD that wave was the busiest of the
    // kernel (it set the step time at B = 32), so the streams from the third on are summed by a SECOND wave, the owner of layer NL-2,
    // which keeps a copy of the last layer's tile for them.
    constexpr bool SPLIT = NS >= 3 && !BIG;
    // (Round 6, measured dead end: a DEDICATED summing wave -- every wave only fills the LDS slots, the wave whose own layers end first
    // spins on them in layer order and publishes -- was 0.75-0.88 us from the last layer's z to the published sum against 0.64 us for the
    // owner of the last layer summing between its polls: 9.0-9.1 instead of 8.5 us per step at batch 8.)
    const int vA = (NL - 1 - NLDS) & 7, vB = (SPLIT && NL >= 2) ? ((NL - 2) & 7) : -1;
    const bool sumA = v == vA, sumB = v == vB, anysum = sumA || sumB;
    Tile wx;                                                      // wave B: the last layer's tile
    float bx = 0.0f;
    if (SPLIT && sumB) {
        const long long lb = L.off_layer0 + (long long)(NL - 1) * L.layer_stride;
        load_tile(wx, a.P + lb + LayerOff::SK + (long long)g * kTile, lane);
        if (use_bias) bx = a.P[lb + LayerOff::SK + (long long)L.NSJ * kTile + g * 64 + lane];
    }
    const int n16 = lane & 15;
    const int zc_lane = lane_of_z((lane < 32 ? 0 : 16) + n16);
    unsigned long long seen[kSlots + 1], period = 0;              // arrival times of the own layers (first stream served) in the previous step
#pragma unroll
    for (int i = 0; i < kSlots + 1; ++i) seen[i] = 0;
    for (int t = 0; t < T && !pl.dead; ++t) {
        asm volatile(".p2align 6");                        // (XALIGN, see chain_many_role)
        const unsigned tag = (unsigned)t + 1u;
        int nextl[NS];
        float tot[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) { nextl[k] = 0; tot[k] = 0.0f; }
        // one own layer: `dot` multiplies by a register tile or (BIG, early layers) by a tile in LDS
        auto own_layer = [&](auto&& dot, const float bias, const int l, const int si, const int k0, const int k1) __attribute__((always_inline)) {
            {
                if (!anysum && period) nap_until(seen[si] + period - (period >> 3));
                // the XCD's streams in a fixed order (they settle a fraction of a microsecond apart).  Measured alternatives: polling all
                // the streams still missing in one round and serving whichever arrived (every extra load of a polling round adds to
                // its round trip: B = 32 ran at 16.5 instead of 13 us/step); requesting the next stream's granules while this stream's
                // dot runs (no gain: 12.0 against 12.1 us/step at B = 32, slower at B = 16).
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    if (pl.dead) break;
                    if (k < k0 || k >= k1) continue;                       // the last layer is served by two waves (see SPLIT)
                    const bool summer = (SPLIT && vB >= 0) ? (k < 2 ? sumA : sumB) : sumA;     // of stream k
                    const rsrc_t rs = sx.rs[k];
                    const int b = sx.b[k];
                    pl.rs = rs;
                    // model.py:154 sum(outputs), in layer order: the values of layers nextl .. l-1 come from the other waves through LDS;
                    // eight slots per LDS round trip (one at a time, the 30 reads of a step cost the wave 1.5 us per stream: with four
                    // streams per XCD that was the step time); consumed strictly in layer order, up to the first one not yet there.
                    // Inside the z poll (non-blocking) the wide form lengthens the polling round: measured on one box, one slot per
                    // round is better up to two streams per XCD (10.47 against 10.67 us/step at B = 8), eight from three on (11.96
                    // against 13.5 us/step at B = 32)
                    constexpr int kNb = NS >= 3 ? 8 : 1;
                    auto drain = [&](bool blocking) __attribute__((always_inline)) {
                        pl.it = 0;
                        while (nextl[k] < l && !pl.dead) {
                            const int base = nextl[k];
                            unsigned long long q[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (blocking || j < kNb) q[j] = LDSU64(k * SW + (base + j < NL ? base + j : NL - 1) * 64 + lane);
                            bool go = true;
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                if (!blocking && j >= kNb) break;
                                go = go && (base + j < l) && __all(g_tag(q[j]) == tag);
                                if (go) { tot[k] = (base + j == 0) ? g_val(q[j]) : tot[k] + g_val(q[j]); nextl[k] = base + j + 1; }
                            }
                            if (nextl[k] == base) {
                                if (!blocking) return;
                                if (!poll_tick(pl, 52)) return;
                            }
                        }
                    };
                    XMARK(ROLE_SKIP0 + g, 10 + si);
                    // ONE load per round (every load of a polling round adds to its round trip): lanes 0-31 fetch z[0..15] twice, lanes
                    // 32-63 z[16..31] twice; v_permlane32_swap makes the two dot operands of it
                    unsigned long long qz;
                    pl.it = 0;
                    // (Round 6, measured dead end: keeping three polls in flight here and two rounds in the conv1 workgroups' poll -- a new load
                    // issued whenever the oldest comes back unsuccessful -- to sample the L2 more often than once per round trip: 8.59 against
                    // 8.52 us per step at batch 8.  Polling loads do not pipeline: the extra loads lengthen every round trip.)
                    for (;;) {
                        qz = xb_load(rs, (int)XcdExch::ZX + l * 128, zc_lane * 2);
                        if (summer) drain(false);                          // while the load is in flight: add what has arrived
                        if (__all(g_tag(qz) == tag)) break;
                        if (!poll_tick(pl, 51)) break;
                        if (!summer) __builtin_amdgcn_s_sleep(1);
                    }
                    if ((INSTR & 1) && pl.dead && g == 0 && lane == 0) {   // bring-up: what the abandoned poll last saw
                        xb_store(rs, (int)XcdExch::MARK + 176 + v * 4, 0, g_tag(qz), __uint_as_float(g_tag(qz)));
                        xb_store(rs, (int)XcdExch::MARK + 177 + v * 4, 0, (unsigned)l, __uint_as_float((unsigned)pl.it));
                        xb_store(rs, (int)XcdExch::MARK + 178 + v * 4, 0, tag, __uint_as_float((unsigned)si));
                    }
                    // (no exit here on a watchdog abort: see the chain role; the loops end at their heads)
                    if (!anysum && k == 0) {
                        const unsigned long long now = __builtin_amdgcn_s_memtime();
                        if (si == 0) { const unsigned long long d = now - seen[0]; period = (seen[0] != 0 && d < (1ull << 18)) ? d : 0; }
                        seen[si] = now;
                    }
                    XSTAMP(g == 0 && summer && l == NL - 1, 20);
                    const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)qz, (unsigned)qz, false, false);   // [lo, lo], [hi, hi]
                    float val = dot(__uint_as_float(sw[0]), __uint_as_float(sw[1]));                             // model.py:96
                    if (use_bias) val = val + bias;
                    if (!summer) {
                        LDSU64(k * SW + l * 64 + lane) = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(val);
                    } else {
                        drain(true);                                       // whatever is still missing below this layer
                        tot[k] = (l == 0) ? val : tot[k] + val;
                        nextl[k] = l + 1;
                        if (l == NL - 1 && !pl.dead) {                     // this stream's sum is complete: out it goes before the next stream's turn
                            const float h = (KF > 0 || tot[k] > 0.0f) ? tot[k] : 0.0f;           // model.py:157 (KF: the sum is not complete yet)
                            xb_store(rs, (int)XcdExch::H1 + g * 64, lane, tag, h);
                            XSTAMP(g == 0, 21);
                        }
                    }
                }
            }
        };
#pragma unroll
        for (int j = 0; j < kLds; ++j) {
            if (j < nlds && !pl.dead)
                own_layer([&](float xa_, float xb_) __attribute__((always_inline)) { return dot32_dpp_lds(o_tile4 + (v + 8 * j) * (kTile / 4), lane, xa_, xb_); }, bs[j], v + 8 * j, j, 0, NS);
        }
#pragma unroll
        for (int i = 0; i < kReg; ++i) {
            if (i < nown && !pl.dead) {
                const int l = NLDS + v + 8 * i;
                own_layer([&](float xa_, float xb_) __attribute__((always_inline)) { return dot32_dpp(ws[i].w, xa_, xb_); }, bs[kLds + i], l, kLds + i, 0,
                          (SPLIT && vB >= 0 && l == NL - 1) ? 2 : NS);
            }
        }
        if (SPLIT && sumB && !pl.dead)
            own_layer([&](float xa_, float xb_) __attribute__((always_inline)) { return dot32_dpp(wx.w, xa_, xb_); }, bx, NL - 1, kSlots, 2, NS);
    }
    if (pl.dead && lane == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) xb_store(sx.rs[k], (int)XcdExch::CTRL + 1, 0, 1u, 0.0f);      // a watchdog on one stream stops them all
    }
}

// =====================================================================================================================
//  CONV1 workgroup g: model.py:158-160 conv1d_1 + relu for output block g (16 chunk tiles over the 8 waves), then the two
//  chunks (2g, 2g+1) of model.py:161-165 conv1d_2 that read this block: what travels on is conv1d_2's partial table.
// =====================================================================================================================
// conv1 workgroups, wave v: the 64 values h1[64 v ..] of a step (relu of model.py:154's skip sum), as one granule per lane.
// KF = 0: skip workgroup v publishes them.  KF = 1 (round 6): skip workgroup v publishes the running sum of layers 0 .. NL-2 (raw); the
// wave holds block v of the LAST layer's skip kernel (wl, bias bl), polls that layer's z (the granule the skip workgroups poll for their
// layers) next to the sum, has the last skip value ready when the sum comes in, and finishes the sum in layer order + relu itself.
template <int KF, bool BAR>
__device__ __forceinline__ unsigned long long conv1_fetch_h1(Poll& pl, const rsrc_t rs, const int v, const int lane, const unsigned tag, const int last_layer,
                                                             const int zc_lane, const float (&wl)[32], const float bl, const bool use_bias, bool& saw_z)
{
    unsigned long long q;
    pl.it = 0;
    if (KF > 0) {
        unsigned long long qz;
        float val = 0.0f;
        bool havez = false;
        for (;;) {
            if (!havez) qz = xb_load_t<BAR>(rs, (int)XcdExch::ZX + last_layer * 128, zc_lane * 2);
            q = xb_load_t<BAR>(rs, (int)XcdExch::H1 + v * 64, lane);
            if (!havez && __all(g_tag(qz) == tag)) {
                const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)qz, (unsigned)qz, false, false);   // [lo, lo], [hi, hi]
                val = dot32_dpp(wl, __uint_as_float(sw[0]), __uint_as_float(sw[1]));                        // model.py:96
                if (use_bias) val = val + bl;
                havez = true;
                saw_z = true;
            }
            if (havez && __all(g_tag(q) == tag)) break;
            if (!poll_tick<BAR>(pl, 61)) break;
            if (!havez) __builtin_amdgcn_s_sleep(1);
        }
        const float tot = g_val(q) + val;                                            // model.py:154: the last term of the sum
        const float h = tot > 0.0f ? tot : 0.0f;                                     // model.py:157
        q = (unsigned long long)__float_as_uint(h);
    } else {
        for (;;) {
            q = xb_load_t<BAR>(rs, (int)XcdExch::H1 + v * 64, lane);
            if (__all(g_tag(q) == tag)) break;
            if (!poll_tick<BAR>(pl, 61)) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return q;
}
// the last layer's skip tile, output block v, for conv1_fetch_h1<1>
__device__ __forceinline__ void conv1_load_last_skip(const XcdLaunch& a, const int v, const int lane, Tile& wl, float& bl)
{
    const Layout& L = a.lay;
    const long long lb = L.off_layer0 + (long long)(L.NL - 1) * L.layer_stride;
    load_tile(wl, a.P + lb + LayerOff::SK + (long long)v * kTile, lane);
    if (L.use_bias != 0) bl = a.P[lb + LayerOff::SK + (long long)L.NSJ * kTile + v * 64 + lane];
}

// KF: see conv1_fetch_h1.  The last layer's z -> skip workgroup -> dot / sum / publish -> conv1 path (0.31 + 0.33 + 0.25 us behind the last
// layer) becomes z -> conv1 (0.31 us) in parallel with the second-to-last layer's trip through the skip workgroup.
template <int INSTR, int NS, bool BAR = false, class SX = XStreams<NS>, int KF = 0>
__device__ __forceinline__ void conv1_role(const XArgs& xa, const SX& sx, int g, const int ns_rt = NS, const int prof_slot = -1)
{
    const XcdLaunch& a = xa.p;
    const Layout& L = a.lay;
    const int v = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int T = a.T, NCH = L.NCH;
    const bool use_bias = L.use_bias != 0;
    Poll pl{sx.rs[0], a.status, 0, false};
    constexpr int O_PART = 0, O_CNT = 16 * 64;                            // LDS floats per stream: chunk partials | arrival counter
    if (threadIdx.x < NS) LDSI(threadIdx.x * kConvLdsFloats + O_CNT) = 0;
    __syncthreads();
    const int c0 = 2 * v, c1 = 2 * v + 1;                                 // this wave's chunks
    Tile ta, tb;
    load_tile(ta, a.P + L.off_w1 + ((long long)g * NCH + c0) * kTile, lane);
    load_tile(tb, a.P + L.off_w1 + ((long long)g * NCH + c1) * kTile, lane);
    const bool summer = v < 2;
    float b1v = 0.0f;
    Tile t2;
    if (summer) {
        if (use_bias) b1v = a.P[L.off_b1 + g * 64 + lane];
        load_tile(t2, a.P + L.off_w2 + (long long)(2 * g + v) * kTile, lane);
    }
    Tile wl;                                                              // KF: the last layer's skip kernel, output block v
    float bl = 0.0f;
    if (KF > 0) conv1_load_last_skip(a, v, lane, wl, bl);
    const int zc_lane = lane_of_z((lane < 32 ? 0 : 16) + (lane & 15));
    unsigned long long t_arr = 0, period = 0;
    WACC_DECL();
    for (int t = 0; t < T && !pl.dead; ++t) {
        asm volatile(".p2align 6");                        // (XALIGN, see chain_many_role)
        const unsigned tag = (unsigned)t + 1u;
        WACC_T0();
        if (period) nap_until(t_arr + period - (period >> 3));
        WACC_T1(0);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (pl.dead) break;
            if (k >= ns_rt) break;                                     // (many-streams kernel: NS = kManyPerXcd = 12 slots, ns_rt streams on this XCD)
            const rsrc_t rs = sx.rs[k];
            const int b = sx.b[k];
            const int ob = k * kConvLdsFloats;
            pl.rs = rs;
            XMARK(ROLE_CONV0 + g, 1);
            // this wave's two chunks are exactly the 64 values skip workgroup v publishes: ONE granule per lane
            WACC_T0();
            bool saw_z = false;
            const unsigned long long q = conv1_fetch_h1<KF, BAR>(pl, rs, v, lane, tag, L.NL - 1, zc_lane, wl.w, bl, use_bias, saw_z);
            (void)saw_z;
            WACC_T1(1);
            WTRACE(prof_slot >= 0 && k == 0, 400 + g * 8 + v);
            // (no exit here on a watchdog abort: see the chain role; the loops end at their heads)
            unsigned long long now_arr = 0;
            if (k == 0) now_arr = __builtin_amdgcn_s_memtime();       // read only: the arithmetic follows the dots (as in the chain)
            XSTAMP(g == 0 && v == 0, 22);
            // rows r0..r3 of the wave = h1[64v + 16r ..]; the dots want "every row r0" / "every row r1" (chunk c0) and r2 / r3 (chunk c1)
            const unsigned hq = (unsigned)q;
            const auto p16 = __builtin_amdgcn_permlane16_swap(hq, hq, false, false);        // [r0,r0,r2,r2], [r1,r1,r3,r3]
            const auto pa = __builtin_amdgcn_permlane32_swap(p16[0], p16[0], false, false);  // [r0 x4], [r2 x4]
            const auto pb = __builtin_amdgcn_permlane32_swap(p16[1], p16[1], false, false);  // [r1 x4], [r3 x4]
            float r0, r1;
            dot32_dpp_x2(ta.w, __uint_as_float(pa[0]), __uint_as_float(pb[0]), tb.w, __uint_as_float(pa[1]), __uint_as_float(pb[1]), r0, r1);
            lds[ob + O_PART + c0 * 64 + lane] = r0;
            lds[ob + O_PART + c1 * 64 + lane] = r1;
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&LDSI(ob + O_CNT), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (k == 0) {
                const unsigned long long d = now_arr - t_arr;
                period = (t_arr != 0 && d < (1ull << 18)) ? d : 0;
                t_arr = now_arr;
            }
            XSTAMP(g == 0 && v == 0, 23);
            XMARK(ROLE_CONV0 + g, 2);
            if (summer) {
                WACC_T0();
                pl.it = 0;
                while (LDSVI(ob + O_CNT) < 8 * (t + 1)) { if (!poll_tick<BAR>(pl, 62)) break; }
                WACC_T1(2);
                // (no exit here on a watchdog abort: see the chain role; the loops end at their heads)
                asm volatile("" ::: "memory");
                XSTAMP(g == 0 && v == 0, 24);
                float cp[16];
#pragma unroll
                for (int ch = 0; ch < 16; ++ch) cp[ch] = lds[ob + O_PART + ch * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
                float r = cp[0];                                            // AC-1: chunk values added in chunk order
#pragma unroll
                for (int ch = 1; ch < 16; ++ch) r = r + cp[ch];
                if (use_bias) r = r + b1v;
                const float h = r > 0.0f ? r : 0.0f;                         // model.py:160
                // conv1d_2 chunk 2g+v reads h[32v .. 32v+31] of this block
                const float p = (v == 0) ? dot_readlane_pipe(t2, h) : dot_readlane_pipe32(t2, h);
                if (lane < 32) xb_store(rs, (int)XcdExch::PT, (g * 32 + lane) * 2 + v, tag, p);     // chunk 2g+v of output `lane`, next to its pair
                WTRACE(prof_slot >= 0 && k == 0, 480 + g * 2 + v);
                XSTAMP(g == 0 && v == 0, 25);
            }
        }
    }
    if (pl.dead && lane == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) xb_store(sx.rs[k], (int)XcdExch::CTRL + 1, 0, 1u, 0.0f);       // the waves waiting on the LDS counter see it in poll_tick
    }
    WACC_OUT(a.prof, prof_slot, v);
}

// =====================================================================================================================
//  CONV1 workgroup g of the ONE-HOT model (256-way softmax output, model.py:161-165 with quantization_channels filters): conv1d_2
//  is 256 x 512 here, sixteen times the MoL head, so it is split by OUTPUT instead of shipping a [16 chunks][256] partial table to
//  the sampler: workgroup g publishes its block of relu(conv1d_1) (64 values), every conv1 workgroup gathers all eight blocks (wave v
//  polls exactly block v: the two chunks 2v, 2v+1 of its conv1d_2 dot, one granule per lane -- the same pattern as the h1 gather),
//  computes the logits of classes 32g .. 32g+31 (lanes 0-31: chunk 2v, lanes 32-63: chunk 2v+1 of the same 32 outputs; ONE 32-term
//  DPP dot per wave), adds the sixteen chunk values in chunk order through LDS (AC-1), then the bias, and publishes 32 logits.
//  One more L2 hop than the MoL head (h2), 256 granules instead of 4096 for the sampler to collect.
// =====================================================================================================================
constexpr int kConvQLdsFloats = 16 * 64 + 64 + 8 * 64;   // LDS floats per stream: conv1d_1 chunk partials | two arrival counters | conv1d_2 chunk partials
template <int INSTR, int NS, int KF = 0>
__device__ __forceinline__ void conv1_onehot_role(const XArgs& xa, const XStreams<NS>& sx, int g)
{
    const XcdLaunch& a = xa.p;
    const Layout& L = a.lay;
    const int v = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int T = a.T, NCH = L.NCH;
    const bool use_bias = L.use_bias != 0;
    Poll pl{sx.rs[0], a.status, 0, false};
    constexpr int O_PART = 0, O_CNT = 16 * 64, O_P2 = 16 * 64 + 64;
    if (threadIdx.x < NS) { LDSI(threadIdx.x * kConvQLdsFloats + O_CNT) = 0; LDSI(threadIdx.x * kConvQLdsFloats + O_CNT + 1) = 0; }
    __syncthreads();
    const int c0 = 2 * v, c1 = 2 * v + 1;                                 // this wave's chunks (of conv1d_1 AND of conv1d_2)
    Tile ta, tb;
    load_tile(ta, a.P + L.off_w1 + ((long long)g * NCH + c0) * kTile, lane);
    load_tile(tb, a.P + L.off_w1 + ((long long)g * NCH + c1) * kTile, lane);
    // conv1d_2: output 32 g + (lane & 31), chunk 2v (lanes 0-31) / 2v+1 (lanes 32-63): column (g & 1) * 32 + (lane & 31) of tile
    // [oblk = g / 2][chunk] in the packed [kq][64][4] layout
    float w2[32];
    {
        const int ch2 = lane < 32 ? c0 : c1;
        const f32x4* src = reinterpret_cast<const f32x4*>(a.P + L.off_w2 + ((long long)(g >> 1) * NCH + ch2) * kTile) + ((g & 1) * 32 + (lane & 31));
#pragma unroll
        for (int q = 0; q < 8; ++q) { const f32x4 w = src[q * 64]; w2[4 * q] = w.x; w2[4 * q + 1] = w.y; w2[4 * q + 2] = w.z; w2[4 * q + 3] = w.w; }
    }
    float b1v = 0.0f, b2v = 0.0f;
    if (use_bias && v == 0) b1v = a.P[L.off_b1 + g * 64 + lane];
    if (use_bias && v == 1 && lane < 32) b2v = a.P[L.off_b2 + g * 32 + lane];
    Tile wl;                                                              // KF: the last layer's skip kernel, output block v (conv1_fetch_h1)
    float bl = 0.0f;
    if (KF > 0) conv1_load_last_skip(a, v, lane, wl, bl);
    const int zc_lane = lane_of_z((lane < 32 ? 0 : 16) + (lane & 15));
    const int cls = g * 32 + (lane & 31);                                 // the class whose logit lanes 0-31 of wave 1 publish
    const int ql_word = (cls & 63) * 4 + (cls >> 6);
    unsigned long long t_arr = 0, period = 0;
    for (int t = 0; t < T && !pl.dead; ++t) {
        asm volatile(".p2align 6");
        const unsigned tag = (unsigned)t + 1u;
        if (period) nap_until(t_arr + period - (period >> 3));
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (pl.dead) break;
            const rsrc_t rs = sx.rs[k];
            const int ob = k * kConvQLdsFloats;
            const int b = sx.b[k];
            (void)b;
            pl.rs = rs;
            // ---- model.py:158-160 conv1d_1 + relu for output block g, as in conv1_role
            bool saw_z = false;
            unsigned long long q = conv1_fetch_h1<KF, false>(pl, rs, v, lane, tag, L.NL - 1, zc_lane, wl.w, bl, use_bias, saw_z);
            (void)saw_z;
            unsigned long long now_arr = 0;
            if (k == 0) now_arr = __builtin_amdgcn_s_memtime();
            XSTAMP(g == 0 && v == 0, 22);
            {
                const unsigned hq = (unsigned)q;
                const auto p16 = __builtin_amdgcn_permlane16_swap(hq, hq, false, false);        // [r0,r0,r2,r2], [r1,r1,r3,r3]
                const auto pa = __builtin_amdgcn_permlane32_swap(p16[0], p16[0], false, false);  // [r0 x4], [r2 x4]
                const auto pb = __builtin_amdgcn_permlane32_swap(p16[1], p16[1], false, false);  // [r1 x4], [r3 x4]
                float r0, r1;
                dot32_dpp_x2(ta.w, __uint_as_float(pa[0]), __uint_as_float(pb[0]), tb.w, __uint_as_float(pa[1]), __uint_as_float(pb[1]), r0, r1);
                lds[ob + O_PART + c0 * 64 + lane] = r0;
                lds[ob + O_PART + c1 * 64 + lane] = r1;
            }
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&LDSI(ob + O_CNT), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (k == 0) {
                const unsigned long long d = now_arr - t_arr;
                period = (t_arr != 0 && d < (1ull << 18)) ? d : 0;
                t_arr = now_arr;
            }
            XSTAMP(g == 0 && v == 0, 23);
            if (v == 0) {
                pl.it = 0;
                while (LDSVI(ob + O_CNT) < 8 * (t + 1)) { if (!poll_tick(pl, 62)) break; }
                asm volatile("" ::: "memory");
                XSTAMP(g == 0, 24);
                float cp[16];
#pragma unroll
                for (int ch = 0; ch < 16; ++ch) cp[ch] = lds[ob + O_PART + ch * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
                float r = cp[0];                                            // AC-1: chunk values added in chunk order
#pragma unroll
                for (int ch = 1; ch < 16; ++ch) r = r + cp[ch];
                if (use_bias) r = r + b1v;
                const float h = r > 0.0f ? r : 0.0f;                         // model.py:160
                xb_store(rs, (int)XcdExch::H2 + g * 64, lane, tag, h);
                XSTAMP(g == 0, 25);
            }
            // ---- model.py:161-165 conv1d_2, classes 32g .. 32g+31: this wave's two chunks read block v of relu(conv1d_1)
            pl.it = 0;
            for (;;) {
                q = xb_load(rs, (int)XcdExch::H2 + v * 64, lane);
                if (__all(g_tag(q) == tag)) break;
                if (!poll_tick(pl, 63)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            XSTAMP(g == 0 && v == 1, 34);
            {
                const unsigned hq = (unsigned)q;
                const auto p16 = __builtin_amdgcn_permlane16_swap(hq, hq, false, false);        // rows 0,1: h[0..15] | h[16..31]; rows 2,3: h[32..47] | h[48..63]
                const float r2 = dot32_dpp(w2, __uint_as_float(p16[0]), __uint_as_float(p16[1]));
                lds[ob + O_P2 + v * 64 + lane] = r2;                         // = [chunk 2v + lane / 32][output lane & 31]
            }
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&LDSI(ob + O_CNT + 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            XSTAMP(g == 0 && v == 1, 35);
            if (v == 1) {
                pl.it = 0;
                while (LDSVI(ob + O_CNT + 1) < 8 * (t + 1)) { if (!poll_tick(pl, 64)) break; }
                asm volatile("" ::: "memory");
                XSTAMP(g == 0, 36);
                float cp[16];
#pragma unroll
                for (int ch = 0; ch < 16; ++ch) cp[ch] = lds[ob + O_P2 + ch * 32 + (lane & 31)];
                __builtin_amdgcn_sched_barrier(0);
                float y = cp[0];                                            // AC-1: chunk values added in chunk order
#pragma unroll
                for (int ch = 1; ch < 16; ++ch) y = y + cp[ch];
                if (use_bias) y = y + b2v;
                if (lane < 32) xb_store(rs, (int)XcdExch::QL, ql_word, tag, y);
                XSTAMP(g == 0, 37);
            }
        }
    }
    if (pl.dead && lane == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) xb_store(sx.rs[k], (int)XcdExch::CTRL + 1, 0, 1u, 0.0f);
    }
}

// =====================================================================================================================
//  LC workgroups: model.py:102-111 create_upsample (row by row) and model.py:75-83 lc_filter|lc_gate of every layer,
//  ahead of the chain through the ring XcdExch::LCR.  Wave gw owns layers gw*lpw .. gw*lpw+lpw-1 (NLC tiles each).
// =====================================================================================================================
template <int INSTR, int NS, bool BIG, bool BAR = false>
__device__ __forceinline__ void lc_role(const XArgs& xa, const XStreams<NS>& sx, int wg, const int ns_rt = NS, const int prof_slot = -1)
{
    const XcdLaunch& a = xa.p;
    const Layout& L = a.lay;
    const int v = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int NL = L.NL, T = a.T, NLC = L.NLC, Lc = L.L, lpw = xa.lc_lpw;
    const int gw = wg * 8 + v;
    const int lfirst = gw * lpw;
    int nown = NL - lfirst;
    nown = nown < 0 ? 0 : (nown > lpw ? lpw : nown);
    if (nown == 0) return;
    Poll pl{sx.rs[0], a.status, 0, false};
    const int* hdr = reinterpret_cast<const int*>(a.cond);
    const int mode = hdr[XH_MODE], rows = hdr[XH_ROWS];
    const float* payload = a.cond + XH_WORDS + (long long)a.B * NL * 64;
    constexpr int kLt = BIG ? 6 : 4;                                      // register tiles per wave (BIG: two layers of three chunks)
    Tile lt[kLt];
#pragma unroll
    for (int idx = 0; idx < kLt; ++idx) {
        if (idx < nown * NLC) {
            const int j = idx / NLC, c = idx - j * NLC;
            load_tile(lt[idx], a.P + L.off_lcw + (long long)(lfirst + j) * L.lcw_stride + (long long)c * kTile, lane);
        }
    }
    // wave-private LDS: two rows of 128 floats (ping-pong of the transposed-conv stages; the final row feeds the dots)
    const int o_row = v * 256;
    const int n16 = lane & 15;
    // upsampling kernels (f_i, 2) and the mixed-radix phase counter of the output row index
    int ph[4] = {0, 0, 0, 0}, frame = 0;
    const int n_up = L.n_up;
    int hop = 1;
    for (int i = 0; i < n_up; ++i) hop *= L.up[i];
    const long long need_rows = (mode == XLC_MEL) ? ((long long)T + hop - 1) / hop : T;
    if (rows < need_rows) {
        if (lane == 0) {
            atomicMax(a.status, 71);
#pragma unroll
            for (int k = 0; k < NS; ++k) xb_store(sx.rs[k], (int)XcdExch::CTRL + 1, 0, 1u, 0.0f);
        }
        return;
    }
    WACC_DECL();
    for (int u = 0; u < T && !pl.dead; ++u) {
        const int t = u;
        // ---- what does not depend on the stream: the taps of this row's phase (model.py:102-111)
        float k0[4] = {0.f, 0.f, 0.f, 0.f}, k1[4] = {0.f, 0.f, 0.f, 0.f};
        if (mode != XLC_UPSAMPLED) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < n_up) { k0[i] = a.P[L.off_up[i] + ph[i] * 2 + 0]; k1[i] = a.P[L.off_up[i] + ph[i] * 2 + 1]; }
        }
        // ---- throttle: slot (u+1) % ring is free once the chain has started step u + 2 - ring (every stream of the XCD)
        WACC_T0();
        if (u + 3 - kXcdLcRing > 0) {
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                if (k >= ns_rt) break;
                const rsrc_t rs = sx.rs[k];
                pl.rs = rs;
                XMARK(ROLE_LC0 + wg, 1);
                pl.it = 0;
                while (!pl.dead) {
                    const unsigned long long q = xb_load_t<BAR>(rs, (int)XcdExch::CTRL, 0);
                    if ((int)g_tag(q) >= u + 3 - kXcdLcRing) break;
                    if (!poll_tick<BAR>(pl, 72)) break;
                    __builtin_amdgcn_s_sleep(16);
                }
            }
            if (pl.dead) break;
        }
        WACC_T1(0);
        // ---- the streams' input rows, all requested before the first is used
        float rowa[NS], rowb[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            rowa[k] = 0.0f; rowb[k] = 0.0f;
            if (k >= ns_rt) continue;
            const float* src = payload + ((long long)sx.b[k] * rows + (mode == XLC_UPSAMPLED ? u : frame)) * Lc;
            rowa[k] = lane < Lc ? src[lane] : 0.0f;
            rowb[k] = 64 + lane < Lc ? src[64 + lane] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (k >= ns_rt) break;
            const rsrc_t rs = sx.rs[k];
            const int b = sx.b[k];
            // ---- row u of the upsampled condition into LDS (zero padded to NLC*32)
            int o_cur = o_row;
            lds[o_cur + lane] = rowa[k];
            lds[o_cur + 64 + lane] = rowb[k];
            if (mode != XLC_UPSAMPLED) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < n_up) {
                        // out[m] = K[a][0]*in[m] + K[a][1]*in[m-1]   as a two-term AC-1 chunk   (model.py:102-111, 'same' transposed conv)
                        const int o_nxt = (o_cur == o_row) ? o_row + 128 : o_row;
#pragma unroll
                        for (int hlf = 0; hlf < 2; ++hlf) {
                            const int m = hlf * 64 + lane;
                            const float x0 = lds[o_cur + m];
                            const float x1 = m > 0 ? lds[o_cur + m - 1] : 0.0f;
                            const float s0 = fma_(k0[i], x0, 0.0f);
                            const float s1 = fma_(k1[i], x1, 0.0f);
                            const float r = (s0 + s1) + (0.0f + 0.0f);
                            lds[o_nxt + m] = m < Lc ? r : 0.0f;
                        }
                        o_cur = o_nxt;
                    }
                }
            }
            // ---- projections: AC-1 chunks of 32 over the lc channels, chunk values added in order
            float resx[kLt];
#pragma unroll
            for (int idx = 0; idx < kLt; ++idx) {
                resx[idx] = 0.0f;
                if (idx < nown * NLC) {
                    const int j = idx / NLC, c = idx - j * NLC;
                    const float xa_ = lds[o_cur + c * 32 + n16], xb_ = lds[o_cur + c * 32 + 16 + n16];
                    resx[idx] = dot32_dpp(lt[idx].w, xa_, xb_);
                    (void)j;
                }
            }
            // combine per layer in chunk order (indices are compile-time; j, c are uniform)
#pragma unroll
            for (int j = 0; j < kLt; ++j) {
                if (j < nown) {
                    float r = 0.0f;
#pragma unroll
                    for (int idx = 0; idx < kLt; ++idx) {
                        if (idx >= j * NLC && idx < (j + 1) * NLC) r = (idx == j * NLC) ? resx[idx] : r + resx[idx];
                    }
                    const int l = lfirst + j;
                    if (u + 1 < T) xb_store(rs, (int)XcdExch::LCR + (((u + 1) % kXcdLcRing) * kXcdLs + l) * 64, lane, (unsigned)u + 2u, r);
                    else {
                        // the frame the NEXT call uses at its step 0.  In a short call nothing has throttled this wave yet, and the service
                        // workgroup may not have read the PREVIOUS call's frame from the same slot: its step-0 addend (PG tag 1) carries that
                        // frame, so the tag says it has been read (found by a one-step call at batch 64: the many-streams service reads its second slot later)
                        if (T <= kXcdLcRing + 1) {
                            pl.rs = rs;
                            pl.it = 0;
                            for (;;) {
                                const unsigned long long ql = xb_load_t<BAR>(rs, (int)XcdExch::PG + l * 64, lane);
                                if (__all(g_tag(ql) >= 1u)) break;
                                if (!poll_tick<BAR>(pl, 73)) break;
                                __builtin_amdgcn_s_sleep(8);
                            }
                        }
                        (a.state + (long long)b * L.state_stride)[L.st_lcprev + l * 64 + lane] = r;
                    }
                }
            }
        }
        if (mode != XLC_UPSAMPLED) {
            // advance the phase counter (last stage fastest)
            int carry = 1;
            for (int i = n_up - 1; i >= 0; --i) {
                if (carry) { ph[i] += 1; carry = 0; if (ph[i] == L.up[i]) { ph[i] = 0; carry = 1; } }
            }
            frame += carry;
        }
    }
    if (pl.dead && lane == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) xb_store(sx.rs[k], (int)XcdExch::CTRL + 1, 0, 1u, 0.0f);
    }
    WACC_OUT(a.prof, prof_slot, v);
}

// ---- co-residency check.  The roles of a launch spin on each other, so ALL of them must be running: with another kernel holding CUs
// (a second model, a Tacotron pass on another stream) some role workgroups stay queued and the rest would poll until the watchdog --
// after having consumed inputs and touched the state.  Every role workgroup therefore counts itself in first and waits, bounded
// (~50 ms), for the rest; the last one to arrive raises GO, the first one to run out of patience raises ABORT (one compare-and-swap
// decides for the whole launch).  On ABORT every workgroup leaves before it has written anything but its ticket: status code 90, the
// state and the outputs are untouched and the caller may simply retry.  Thread 0 also checks that the conditioning buffer carries
// the header twv_wavenet_condition[_mel] writes for THIS kernel (code 74: a buffer built while the generic kernel was selected).
// roles[0..7] per-XCD tickets, roles[8] arrivals, roles[9] decision (0 undecided, 1 go, 2 abort).
__device__ __forceinline__ bool roles_resident(const XArgs& xa)
{
    const XcdLaunch& a = xa.p;
    __shared__ int s_go;
    if (threadIdx.x == 0) {
        int* arrived = a.roles + 8;
        int* decision = a.roles + 9;
        const int* hdr = reinterpret_cast<const int*>(a.cond);
        if (hdr[XH_MAGIC] != kXcdCondMagic || (unsigned)hdr[XH_MODE] > (unsigned)XLC_MEL) {
            atomicMax(a.status, 74);
            atomicCAS(decision, 0, 2);
        }
        if (atomicAdd(arrived, 1) + 1 == xa.total_roles) atomicCAS(decision, 0, 1);
        int f = 0;
#pragma nounroll
        for (int i = 0; i < (1 << 15); ++i) {
            f = atomicAdd(decision, 0);
            if (f != 0) break;
            __builtin_amdgcn_s_sleep(32);
        }
        if (f == 0) {
            const int old = atomicCAS(decision, 0, 2);
            f = old == 0 ? 2 : old;
            if (old == 0) atomicMax(a.status, 90);
        }
        s_go = (f == 1) ? 1 : 0;
    }
    __syncthreads();
    return s_go != 0;
}

// ONE: see xcd_launch.  BIGK: the instantiation for 31-50 layers (second chain workgroup, helper waves with LDS-resident early tiles).  It is a kernel of
// its own so that the register allocation of the 30-layer kernel is not touched by it (as ONE kernel the sampling loop of the bench
// configuration ran at 13.3 instead of 10.4 us per step).
// ONEHOT: the one-hot mu-law model (scalar_input False, 256 classes): its own kernel instantiations, the MoL kernels carry none of it.
template <int INSTR, bool BIGK, bool ONE, bool ONEHOT = false>
__global__ void __launch_bounds__(512) wn_xcd_generate_kernel(XArgs xa)
{
    const XcdLaunch& a = xa.p;
    // ---- role ticket of this workgroup's XCD (HIP promises nothing about workgroup -> XCD placement: ask the hardware)
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    int ticket = 0;
    if (threadIdx.x == 0) ticket = (xcc < (unsigned)a.B && xcc < 8u) ? atomicAdd(a.roles + xcc, 1) : 1 << 20;
    ticket = __shfl(ticket, 0);
    __shared__ int s_ticket;
    if (threadIdx.x == 0) s_ticket = ticket;
    // LDS hand-off words start at zero
    for (int i = threadIdx.x; i < kXcdStreamsPerXcd * kSkipLdsWords * 2 + 64; i += blockDim.x) lds[i] = 0.0f;
    __syncthreads();
    ticket = s_ticket;
    // roles of an XCD with ns streams: ns chains (of one or two workgroups), ns service workgroups, then ONE set of 8 skip + 8 conv1 +
    // n_lc workgroups
    const int ns = xcc < (unsigned)a.B && xcc < 8u ? (a.B - (int)xcc + 7) / 8 : 0;
    constexpr int nseg = BIGK ? 2 : 1;
    const int nchain = ns * nseg;
    if (ticket >= nchain + ns + 16 + xa.n_lc_wg) return;        // surplus workgroup (or an XCD without a stream)
    if (!roles_resident(xa)) return;                            // the device is busy: nobody starts (status 90)
    const bool forced = a.forced != nullptr;
    auto exch_of = [&](int b) { return __builtin_amdgcn_make_buffer_rsrc(a.exch + (long long)b * XcdExch::WORDS, 0, (int)(XcdExch::WORDS * 8), 0x00020000); };
    if (ticket < nchain + ns) {
        const int k = ticket < nchain ? ticket / nseg : ticket - nchain;
        const int b = (int)xcc + 8 * k;
        const rsrc_t rs = exch_of(b);
        if (ticket < nchain) {
            // (AC-1b: bias / gc / lc reach the chain inside the service workgroup's addend: the chain code does not depend on which of
            // them the model has, every model takes the instantiations with the layer count folded)
            if (ticket % nseg == 0) {
                // the sampling chain of the hparams-default model once per layer count of a wave (first chain workgroup: 4 or 3)
                const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
                const int nlw = a.lay.NL - (wv < 6 ? 4 * wv : 24 + 3 * (wv - 6));
                if (forced) chain_role<INSTR, true, true, false, BIGK, -1, ONEHOT>(xa, b, rs);
                else if (nlw >= 4 && wv < 6) chain_role<INSTR, true, false, false, BIGK, 4, ONEHOT, 0>(xa, b, rs);
                else if (nlw >= 3 && wv == 6) chain_role<INSTR, true, false, false, BIGK, 3, ONEHOT, 0>(xa, b, rs);
                else if (nlw >= 3 && wv == 7) chain_role<INSTR, true, false, false, BIGK, 3, ONEHOT, 1>(xa, b, rs);
                else chain_role<INSTR, true, false, false, BIGK, -1, ONEHOT>(xa, b, rs);
            } else if constexpr (BIGK) {
                // (the second chain workgroup has no head: nothing of it depends on the input / output type)
                const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
                if (forced) chain_role<INSTR, true, true, true, true, -1>(xa, b, rs);
                else if (a.lay.NL - kXcdSeg0Layers - 4 * wv >= 4) chain_role<INSTR, true, false, true, true, 4>(xa, b, rs);
                else chain_role<INSTR, true, false, true, true, -1>(xa, b, rs);
            }
        }
        else service_role<INSTR, BIGK>(xa, b, rs);
        return;
    }
    const int role = ticket - nchain - ns;                       // 0-7 skip, 8-15 conv1, 16.. lc
    auto shared_roles = [&](auto nsc) {
        constexpr int NS = decltype(nsc)::value;
        XStreams<NS> sx;
#pragma unroll
        for (int k = 0; k < NS; ++k) { sx.b[k] = (int)xcc + 8 * k; sx.rs[k] = exch_of(sx.b[k]); }
        // the last layer's skip 1x1 runs in the conv1 workgroups (KF = 1; needs a layer in front of it)
        // The unfolded form (a model of KF layers or fewer) is only compiled into the production builds: in the instrumented ones (layer
        // dumps, phase stamps: every stream count in ONE kernel) the second copy of the skip / conv1 roles pushed the register allocation
        // of the whole kernel into scratch (720 B per lane: the stamped chain ran 13 % slower than the one it is meant to show);
        // xcd_launch refuses such a model for those builds.
        constexpr int KF = TWV_XCD_KF;
        constexpr bool kUnfolded = KF == 0 || INSTR == 0;
        const bool fold = KF > 0 && (a.lay.NL > KF || !kUnfolded);
        if (role < 8) {
            if (!forced) {
                if (fold) skip_role<INSTR, NS, BIGK, KF>(xa, sx, role);
                else if constexpr (kUnfolded) skip_role<INSTR, NS, BIGK, 0>(xa, sx, role);
            }
        }
        else if (role < 16) {
            if (!forced) {
                if constexpr (ONEHOT) {
                    if (fold) conv1_onehot_role<INSTR, NS, KF>(xa, sx, role - 8);
                    else if constexpr (kUnfolded) conv1_onehot_role<INSTR, NS, 0>(xa, sx, role - 8);
                }
                else if (fold) conv1_role<INSTR, NS, false, XStreams<NS>, KF>(xa, sx, role - 8);
                else if constexpr (kUnfolded) conv1_role<INSTR, NS>(xa, sx, role - 8);
            }
        }
        else lc_role<INSTR, NS, BIGK>(xa, sx, role - 16);
    };
    if constexpr (BIGK) {                                         // at most two streams per XCD (LDS of the skip workgroups)
        if (ns == 1) shared_roles(std::integral_constant<int, 1>{});
        else shared_roles(std::integral_constant<int, 2>{});
    } else if constexpr (ONE) {                                   // batch <= 8 (see xcd_launch)
        shared_roles(std::integral_constant<int, 1>{});
    } else {
        if (ns == 1) shared_roles(std::integral_constant<int, 1>{});
        else if (ns == 2) shared_roles(std::integral_constant<int, 2>{});
        else if (ns == 3) shared_roles(std::integral_constant<int, 3>{});
        else shared_roles(std::integral_constant<int, 4>{});
    }
}

// =====================================================================================================================
//  THE MANY-STREAMS KERNEL (batch 33 .. 96: five to twelve streams per XCD, two per chain workgroup).
//
//  The chain workgroup's weights serve a stream ~1 us of every ~9.5 us step per wave (SQ_WAIT_ANY 0.83 at batch 8), so TWO streams
//  rotate through one chain workgroup here (slots 0 / 1 = streams 2c / 2c + 1 of the XCD): wave w runs slot 0's layers, hands
//  the residual vector on, runs slot 1's layers; register-resident kernels shared, per-slot state = the hand-off boxes, the gc
//  projections (LDS) and, on wave 7, the causal queue and the sampler's noise terms.  Wave 7 draws a slot's sample and feeds its
//  causal layer in one go (sampler(t-1) -> head(t) per slot, then the slots' layers): neither stream waits for the other's post phase.
//  The service workgroup carries the same two streams (tap-0 kernels shared, delay lines per stream).  Eight streams: 4 chain + 4
//  service + 8 skip + 8 conv1 + 4 lc workgroups = 28 of the XCD's 32 CUs; TWELVE streams (batch 96, kManyChains = 6): 6 + 6 + 8 + 8 + 4 =
//  all 32 CUs carry a role -- at 89 .. 96 streams there is no CU to spare, any other resident kernel ends the launch with TWV_E_BUSY
//  (nothing done; the Python generate() retries a few times).
//
//  The skip role is laid out differently: with eight streams the 32 (layer, stream) polls per wave and step of skip_role (a wave
//  owns slice g of FOUR layers) are an L2 round trip each -- more than the step.  Here a wave owns FOUR SLICES of ONE layer: one
//  poll per stream and step, four dots.  Workgroup (q, hh) = layer group q (eight consecutive layers, the first group short so that
//  the last one is full) x output half hh (256 columns); model.py:154's sum over the layers IN LAYER ORDER runs as a relay through
//  the group's waves (tagged LDS words, like the chain's hand-off) and on to the next group through one L2 granule hop, which is
//  off the sample path: the total of layers 0..l-1 is waiting when layer l's value appears.  Same adds in the same order: same bits.
// =====================================================================================================================
// where a wave is (layer-dump build only): {stage, step + 1} in the MARK area of the stream it is working for -- read back by
// scripts/many_check.py after a watchdog abort.  Slots: 0-7 chain waves, 8-15 service waves, 16 + 8 r + m skip, 80 + 8 g + v conv1
#define MMARK(rs_, slot_, stage_) do { if ((INSTR & 2) && lane == 0) xb_store(rs_, (int)XcdExch::MARK + (slot_), 0, (unsigned)t + 1u, (float)(stage_)); } while (0)
constexpr int kMS = 2;                                   // stream slots of a chain / service workgroup
constexpr int kManyChains = 6;                           // chain (and service) workgroups per XCD: 6 + 6 + 8 skip + 8 conv1 + 4 lc = all 32 CUs at twelve streams
constexpr int kManyPerXcd = kMS * kManyChains;           // streams per XCD
constexpr int kM_BOX = 10;                               // hand-off boxes per slot: 0..7 the waves' inputs, 8 end of a forced step, 9 sink
constexpr int kM_OWD = 4096;                             // LDS floats: dense kernels [30 layers][4][64 lanes][4] behind the boxes
constexpr int kM_OGC = kM_OWD + kXcdSeg0Layers * 1024;   // gc projections [slot][30][64]
constexpr int kM_OHS = kM_OGC + kMS * kXcdSeg0Layers * 64;                 // wave 7's per-slot state [slot][9][64]: causal queue (2), partial chunk (4), noise terms (2), first input
constexpr int kManyChainLds = kM_OHS + kMS * 9 * 64;                       // floats (155.5 KiB)
// skip role: the relay's mailboxes [stream][wave]: the four slices' running totals as one 16-byte word per lane + one tag word per lane
// (a wave's LDS writes are performed in order: data, then tag; the reader asks for the tag, then the data)
constexpr int kManySkipData4 = kManyPerXcd * 8 * 64;                       // float4 units
constexpr int kManySkipLds = kManySkipData4 * 4 + kManyPerXcd * 8 * 64;    // floats (120 KiB at twelve streams)
constexpr int kManyLds = kManyChainLds > kManySkipLds ? kManyChainLds : kManySkipLds;


// mixture.py:84-114 from conv1d_2's [16 chunks][32 outputs] partial table of ONE stream (the sampler half of chain_role, as a function:
// the many-streams chain draws for two slots).  Returns the sample; `tag` = step + 1 of the step the table belongs to.
template <int INSTR>
__device__ __forceinline__ float many_sample(const XcdLaunch& a, const Layout& L, Poll& pl, rsrc_t rs, int lane, unsigned tag, float b2v,
                                             float s_lnl, float s_tq, bool use_bias, int b, int NL)
{
    const int half = lane >> 5;
    unsigned long long q[8];
    pl.it = 0;
    for (;;) {
        bool good = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32x4s d = xbm_load2(rs, (int)XcdExch::PT, (half * 4 + k) * 32 + (lane & 31));
            q[2 * k] = ((unsigned long long)d.y << 32) | d.x;
            q[2 * k + 1] = ((unsigned long long)d.w << 32) | d.z;
            good = good && d.y == tag && d.w == tag;
        }
        if (__all(good)) break;
        if (!poll_tick<true>(pl, 34)) break;
    }
    if ((INSTR & 2) && pl.dead) {                              // bring-up: what the abandoned poll last saw (lanes 0 and 32)
        if ((lane & 31) == 0) {
            xb_store(rs, (int)XcdExch::MARK + 240 + (lane >> 5) * 4, 0, tag, __uint_as_float(g_tag(q[0])));
            xb_store(rs, (int)XcdExch::MARK + 241 + (lane >> 5) * 4, 0, (unsigned)pl.it, __uint_as_float(g_tag(q[7])));
        }
    }
    float y = g_val(q[0]);                                     // chunk partials added in chunk order (AC-1)
#pragma unroll
    for (int k = 1; k < 8; ++k) y = y + g_val(q[k]);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)q[k], (unsigned)q[k], false, false);
        y = y + __uint_as_float(sw[1]);                        // chunk 8 + k of the same output (upper half-wave's granule)
    }
    if (use_bias && lane < L.O) y = y + b2v;
    if ((INSTR & 2) && a.dbg != nullptr && (int)tag - 1 < a.dbg_steps)
        a.dbg[((long long)b * a.dbg_steps + ((int)tag - 1)) * ((long long)NL * 64 + L.Opad) + (long long)NL * 64 + lane] = y;
    const int nr = L.nr_mix;
    const float lsmin = (float)-32.23619130191664;             // mixture.py:107 exp(max(log_scale, log 1e-14)) on every lane
    const float e_all = exp_e(y > lsmin ? y : lsmin);
    const float ninf = __uint_as_float(0xff800000u);           // mixture.py:103 argmax_i(logit_i - log(-log u_i)), first maximum
    const float gmb = (lane < nr) ? y - s_lnl : ninf;
    float mx = gmb;
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ninf), __float_as_int(mx), 0x111, 0xf, 0xf, false)));
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ninf), __float_as_int(mx), 0x112, 0xf, 0xf, false)));
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ninf), __float_as_int(mx), 0x114, 0xf, 0xf, false)));
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ninf), __float_as_int(mx), 0x118, 0xf, 0xf, false)));
    const float best = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mx), 15));
    const unsigned long long hit = __ballot(lane < nr && gmb == best);
    const int k = hit ? (int)__ffsll((long long)hit) - 1 : 0;
    const float mean = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), nr + k));        // mixture.py:105
    const float e = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e_all), 2 * nr + k));    // mixture.py:107
    const float prod = e * s_tq;                               // mixture.py:110-111
    float xs = mean + prod;
    xs = xs > -1.0f ? xs : -1.0f;                              // mixture.py:113
    xs = xs < 1.0f ? xs : 1.0f;
    return xs;
}

// ---- CHAIN workgroup c of an XCD, two stream slots (model.py:41-46, 66-101; mixture.py:84-114 on wave 7) ----------------------
// The slot loop is a real loop (one copy of the layer code, as in chain_role): per-slot values are derived from the slot number
// (stream, buffer descriptor) or live in LDS rows between a slot's turns (wave 7's causal queue / noise terms, the gc projections).
template <int INSTR, bool ALL, bool FORCED, int NC>
__device__ __forceinline__ void chain_many_role(const XArgs& xa, int xcc, int ns, int c)
{
    const XcdLaunch& a = xa.p;
    const Layout& L = a.lay;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int NL = L.NL, T = a.T;
    const bool use_bias = L.use_bias != 0, has_gc = L.G > 0, has_lc = L.L > 0;
    constexpr bool forced = FORCED;
    const int l0 = w < 6 ? 4 * w : 24 + 3 * (w - 6);
    const int cap = w < 6 ? 4 : 3;
    int nl = NL - l0;
    nl = nl < 0 ? 0 : (nl > cap ? cap : nl);
    const int nlc = NC >= 0 ? NC : nl;
    const bool next_has = (l0 + nl < NL);
    const bool head = (w == 7);                                // sampler + causal layer
    if (nl == 0 && !head) return;
    const int oc = dpp_conv_out(lane), od = dpp_dense_out(lane);
    const ActCoef coef = act_coef(lane >= 32);
    // slot s carries stream k = 2 c + s of this XCD: chain-mates are NEIGHBOURS in the order the shared workgroups serve the XCD's streams
    // (k = 0, 1, ..): slot 1 follows slot 0 through the waves about one wave-time (1 us) later, which is also the spacing that order
    // settles into.  (With k = c and c + 4 in one workgroup the shared roles held the mates >= 4 service times apart, wave 7 was still
    // busy with slot 1's layers when slot 0's next sample was due, and one extra stream cost every stream of the XCD 0.9 us per step.)
    const int nslot = (2 * c + 1 < ns) ? 2 : 1;
    auto stream_of = [&](int s) { return xcc + 8 * (2 * c + s); };
    Poll pl{exch_rsrc(a, stream_of(0)), a.status, 0, false};

    // ---- the wave's layers: tap-1 conv kernel register-resident for the whole launch, dense kernel and gc projections in LDS
    ChainRegs W[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < nl) {
            const f32x4* src = reinterpret_cast<const f32x4*>(a.P + L.off_xl + (long long)(l0 + i) * kXcdXlFloats) + lane;
#pragma unroll
            for (int q = 0; q < 8; ++q) { const f32x4 v = src[q * 64]; W[i].wc[4 * q] = v.x; W[i].wc[4 * q + 1] = v.y; W[i].wc[4 * q + 2] = v.z; W[i].wc[4 * q + 3] = v.w; }
#pragma unroll
            for (int q = 0; q < 4; ++q) LDS4((kM_OWD >> 2) + ((l0 + i) * 4 + q) * 64 + lane) = src[(8 + q) * 64];
            const f32x4 v = src[12 * 64];
            W[i].bd = dense_bias_init(lane, use_bias ? v.y : 0.0f);
        }
    }
    // wave 7's per-slot state lives in LDS rows (kM_OHS) between a slot's turns: rows 0 ha, 1 hb (the causal queue, model.py:52, as two
    // row-broadcast registers), 2-5 the causal chunk without its newest term, 6 / 7 the sampler's noise terms, 8 the first input
#define hs(s_, row_) lds[kM_OHS + ((s_) * 9 + (row_)) * 64 + lane]      /* indexed on the LDS symbol itself: a float& would make these FLAT accesses */
    float b2v = 0.0f;
    const bool sampler = head && !forced;
    const bool is15 = (lane & 15) == 15;
    auto causal_prepare = [&](float& ha_, float& hb_, float (&cp_)[4]) __attribute__((always_inline)) {
        const float t1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ha_), 0x101, 0xf, 0xf, true));   // row_shl:1
        const float b0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(hb_), 0x150, 0xf, 0xf, true));   // row_newbcast:0
        ha_ = is15 ? b0 : t1;
        hb_ = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(hb_), 0x101, 0xf, 0xf, true));
        causal_partial_dpp(W[3].wc, ha_, hb_, cp_);
    };
    if (head) {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.P + L.off_xc) + lane;
#pragma unroll
        for (int q = 0; q < 8; ++q) { const f32x4 v = src[q * 64]; W[3].wc[4 * q] = v.x; W[3].wc[4 * q + 1] = v.y; W[3].wc[4 * q + 2] = v.z; W[3].wc[4 * q + 3] = v.w; }
        if (sampler && use_bias && lane < L.O) b2v = a.P[L.off_b2 + lane];
        for (int s = 0; s < nslot; ++s) {
            const float* stb = a.state + (long long)stream_of(s) * L.state_stride;
            float ha_ = stb[L.st_hist + (lane & 15)];
            float hb_ = stb[L.st_hist + 16 + (lane & 15)];
            float cp_[4];
            causal_prepare(ha_, hb_, cp_);
            hs(s, 0) = ha_; hs(s, 1) = hb_; hs(s, 2) = cp_[0]; hs(s, 3) = cp_[1]; hs(s, 4) = cp_[2]; hs(s, 5) = cp_[3];
            hs(s, 6) = 0.0f; hs(s, 7) = 0.0f;
            hs(s, 8) = forced ? 0.0f : reinterpret_cast<const float*>(a.first_input)[stream_of(s)];
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    asm volatile(".p2align 6" ::: "memory");

    unsigned long long tin0 = 0, tin1 = 0, per0 = 0, per1 = 0;  // per slot: when the wave's input arrived in the previous step, the step period
    WACC_DECL();
    for (int t = 0; t < T && !pl.dead; ++t) {
        asm volatile(".p2align 6");                        // the step loop starts on a 64-byte line whatever was compiled in front of it (XALIGN)
        const unsigned tag = (unsigned)t + 1u;
        // ---- wave 7: per slot, the sample of step t-1 straight into the causal layer of step t (one fma and three adds on the
        // sample-to-sample path), result to wave 0.  Slot after slot: neither stream waits for the other's post phase.
        if (head) {
#pragma nounroll
            for (int s = 0; s < nslot && !pl.dead; ++s) {
                const int b = stream_of(s);
                const rsrc_t rs = exch_rsrc(a, b);
                pl.rs = rs;
                float ha_ = hs(s, 0), hb_ = hs(s, 1), cp_[4] = {hs(s, 2), hs(s, 3), hs(s, 4), hs(s, 5)};      // requested before the wait below
                const float lnl_ = hs(s, 6), tq_ = hs(s, 7);
                float s_in = hs(s, 8);
                MMARK(rs, w, 1);
                if (forced) {
                    if (t > 0) {                                    // teacher-forced: the end of step t-1 (box 8) releases step t
                        pl.it = 0;
                        for (;;) {
                            const unsigned long long q = LDSU64((s * kM_BOX + 8) * 64 + lane);
                            if (__all(g_tag(q) == (unsigned)t)) break;
                            if (!poll_tick<true>(pl, 35)) break;
                            __builtin_amdgcn_s_sleep(2);
                        }
                    }
                    s_in = reinterpret_cast<const float*>(a.forced)[(long long)b * T + t];
                } else if (t > 0) {
                    __builtin_amdgcn_s_setprio(3);
                    WACC_T0();
                    s_in = many_sample<INSTR>(a, L, pl, rs, lane, (unsigned)t, b2v, lnl_, tq_, use_bias, b, NL);
                    WACC_T1(2);                                  // (poll + the sampler's arithmetic)
                }
                // (a watchdog abort ends the loops at their heads)
                MMARK(rs, w, 2);
                const float c3 = fma_(W[3].wc[31], s_in, cp_[3]);            // k = 31, the last term of chain 3
                const float x0 = (cp_[0] + cp_[1]) + (cp_[2] + c3);          // model.py:41-46: one AC-1 chunk, no bias; X layout
                LDSU64((s * kM_BOX + 0) * 64 + lane) = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(x0);
                WTRACE(xcc == 0 && c == 0 && s == 0, 0);
                if ((INSTR & 4) && a.prof != nullptr && t == kTraceStep + 1 && xcc == 0 && c == 0 && s == 0 && lane == 0) a.prof[2048 + 1] = wall_clock64();   // the next step's push
                if (lane == 0) {
                    if (sampler && t > 0) a.out[(long long)b * T + t - 1] = s_in;
                    xb_store(rs, (int)XcdExch::CTRL, 0, tag, 0.0f);          // step t has started (the lc workgroups throttle on it)
                }
                hb_ = is15 ? s_in : hb_;                                     // (ha, hb) = the queue after step t
                if (t + 1 < T) causal_prepare(ha_, hb_, cp_);
                hs(s, 0) = ha_; hs(s, 1) = hb_; hs(s, 2) = cp_[0]; hs(s, 3) = cp_[1]; hs(s, 4) = cp_[2]; hs(s, 5) = cp_[3];
                if (sampler) {
                    // mixture.py:103 -log(-log u) per mixture lane; mixture.py:110-111 log u - log(1 - u) of the last draw
                    const float* up = a.uniforms + ((long long)b * T + t) * (L.nr_mix + 1);
                    const float u = lane <= L.nr_mix ? up[lane] : 0.5f;
                    hs(s, 6) = log_e(-log_e(u));
                    const float uu = __shfl(u, L.nr_mix);
                    hs(s, 7) = log_e(uu) - log_e(1.0f - uu);
                }
                __builtin_amdgcn_s_setprio(0);
            }
        }
        // ---- the wave's layers, slot after slot
        if (nlc > 0) {
#pragma nounroll
            for (int s = 0; s < nslot && !pl.dead; ++s) {
                const int b = stream_of(s);
                const rsrc_t rs = exch_rsrc(a, b);
                pl.rs = rs;
                float pre[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                MMARK(rs, w, 3);
                WACC_T0();
                // (A) this step's addends of the wave's layers: ((tap-0 chunk + bias) + gc) + lc (service workgroup; long since published)
                pl.it = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (i < nlc) {
                            const unsigned long long qp = xbm_load(rs, (int)XcdExch::PG + (l0 + i) * 64, oc);
                            ok = ok && g_tag(qp) == tag;
                            pre[i] = g_val(qp);
                        }
                    }
                    if (__all(ok)) break;
                    if (!poll_tick<true>(pl, 31)) break;
                    __builtin_amdgcn_s_sleep(4);
                }
                MMARK(rs, w, 4);
                WACC_T1(0);
                WACC_T0();
                // (B) the wave's input: the previous wave's residual vector (wave 0: the causal layer's output); sleep until its turn is near
                const unsigned long long t_in = s ? tin1 : tin0, in_period = s ? per1 : per0;
                if (in_period) nap_until(t_in + in_period - (in_period >> 4));
                unsigned long long q;
                pl.it = 0;
                for (;;) {
                    q = LDSU64((s * kM_BOX + w) * 64 + lane);
                    if (__all(g_tag(q) == tag)) break;
                    if (!poll_tick<true>(pl, 33)) break;
                }
                __builtin_amdgcn_s_setprio(3);
                float X = g_val(q);
                const unsigned long long now_in = __builtin_amdgcn_s_memtime();
                WACC_T1(1);
                auto run_layers = [&](auto nc) __attribute__((always_inline)) {
                    constexpr int N = decltype(nc)::value;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        float wd[16];
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const f32x4 v = LDS4((kM_OWD >> 2) + ((l0 + i) * 4 + qq) * 64 + lane);
                            wd[4 * qq] = v.x; wd[4 * qq + 1] = v.y; wd[4 * qq + 2] = v.z; wd[4 * qq + 3] = v.w;
                        }
                        const float z = layer_front_dpp(W[i].wc, coef, X, pre[i]);
                        xb_store2(rs, (int)XcdExch::ZX + (l0 + i) * 128, lane, tag, z, X);     // {z, tag} -> skip, {layer input, tag} -> service
                        WTRACE(xcc == 0 && c == 0 && s == 0, 64 + l0 + i);
                        layer_back_dpp(wd, W[i].bd, z, X);
                        if ((INSTR & 2) && a.dbg != nullptr && t < a.dbg_steps) {
                            float* dp = a.dbg + ((long long)b * a.dbg_steps + t) * ((long long)NL * 64 + L.Opad) + (long long)(l0 + i) * 64;
                            if (lane < 32) dp[dpp_z_index(lane)] = z;
                            if ((lane & 16) == 0) dp[32 + od] = X;
                        }
                    }
                };
                if constexpr (NC > 0) run_layers(std::integral_constant<int, NC>{});
                else {
                    if (nl == 4) run_layers(std::integral_constant<int, 4>{});
                    else if (nl == 3) run_layers(std::integral_constant<int, 3>{});
                    else if (nl == 2) run_layers(std::integral_constant<int, 2>{});
                    else if (nl == 1) run_layers(std::integral_constant<int, 1>{});
                }
                LDSU64((s * kM_BOX + (next_has ? w + 1 : 9)) * 64 + lane) = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(X);
                if (forced && !next_has) LDSU64((s * kM_BOX + 8) * 64 + lane) = (unsigned long long)tag << 32;    // the stack is through with step t
                const unsigned long long d = now_in - t_in;
                const unsigned long long np = (t_in != 0 && d < (1ull << 18)) ? d : 0;
                if (s) { per1 = np; tin1 = now_in; } else { per0 = np; tin0 = now_in; }
                __builtin_amdgcn_s_setprio(0);
                MMARK(rs, w, 5);
            }
        }
    }
    // ---- the last sample; persist the causal queue (model.py:49-64), canonical order
    if (head) {
        for (int s = 0; s < nslot; ++s) {
            const int b = stream_of(s);
            float* stb = a.state + (long long)b * L.state_stride;
            if (sampler && !pl.dead && T > 0) {
                const rsrc_t rs = exch_rsrc(a, b);
                pl.rs = rs;
                const float last = many_sample<INSTR>(a, L, pl, rs, lane, (unsigned)T, b2v, hs(s, 6), hs(s, 7), use_bias, b, NL);
                if (lane == 0 && !pl.dead) a.out[(long long)b * T + T - 1] = last;
            }
            if (lane < 16) stb[L.st_hist + lane] = hs(s, 0);
            else if (lane < 32) stb[L.st_hist + lane] = hs(s, 1);
            if (lane == 0) {
                int* meta = reinterpret_cast<int*>(stb + L.st_meta);
                meta[M_TABS] = meta[M_TABS] + T;
            }
        }
    }
    if (pl.dead && lane == 0)
        for (int s = 0; s < nslot; ++s) xb_store(exch_rsrc(a, stream_of(s)), (int)XcdExch::CTRL + 1, 0, 1u, 0.0f);
    WACC_OUT(a.prof, xcc == 0 ? c : -1, w);
}
#undef hs

// ---- SERVICE workgroup c of an XCD, two stream slots: delay lines + tap-0 chunks one step ahead (service_role for two streams) ---
template <int INSTR>
__device__ __forceinline__ void service_many_role(const XArgs& xa, int xcc, int ns, int c)
{
    const XcdLaunch& a = xa.p;
    const Layout& L = a.lay;
    const int sv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int NL = L.NL, T = a.T;
    const bool has_lc = L.L > 0, use_bias = L.use_bias != 0, has_gc = L.G > 0;
    const int* pmeta = reinterpret_cast<const int*>(a.P + L.off_meta);
    constexpr int kReg = 4;
    int nown = 0;
#pragma unroll
    for (int i = 0; i < kReg; ++i) if (sv + 8 * i < NL) nown = i + 1;
    if (nown == 0) return;
    bool ex[kMS];
    rsrc_t rsv[kMS];
    float* stb[kMS];
    int bof[kMS];
#pragma unroll
    for (int j = 0; j < kMS; ++j) {
        const int k = 2 * c + j;                               // the chain workgroup's two streams
        ex[j] = k < ns;
        const int b = ex[j] ? xcc + 8 * k : xcc + 8 * 2 * c;
        bof[j] = b;
        rsv[j] = exch_rsrc(a, b);
        stb[j] = a.state + (long long)b * L.state_stride;
    }
    Poll pl{rsv[0], a.status, 0, false};
    WACC_DECL();
    Tile t0[kReg];
    unsigned dil[kReg], roff[kReg], pos0[kReg][kMS];
    float bfg[kReg], gcv[kReg][kMS];         // conv bias and the hoisted gc projection (per stream) of this lane's conv output
#pragma unroll
    for (int i = 0; i < kReg; ++i) {
        dil[i] = 1; roff[i] = 0; bfg[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < kMS; ++j) { pos0[i][j] = 0; gcv[i][j] = 0.0f; }
        if (i < nown) {
            const int l = sv + 8 * i;
            load_tile(t0[i], a.P + L.off_layer0 + (long long)l * L.layer_stride + LayerOff::T0, lane);
            dil[i] = (unsigned)pmeta[l]; roff[i] = (unsigned)pmeta[64 + l];
            if (use_bias) bfg[i] = a.P[L.off_layer0 + (long long)l * L.layer_stride + LayerOff::BFG + lane];
#pragma unroll
            for (int j = 0; j < kMS; ++j) {
                if (!ex[j]) continue;
                pos0[i][j] = (unsigned)reinterpret_cast<const int*>(stb[j] + L.st_ringpos)[l];
                if (has_gc) gcv[i][j] = a.cond[XH_WORDS + ((long long)bof[j] * NL + l) * 64 + lane];
            }
        }
    }
    // AC-1b addend (see service_role): ((tap-0 chunk + bias) + gc) + lc
    auto addend = [&](int i, int j, float pre, float lcv) __attribute__((always_inline)) -> float {
        float v = pre;
        if (use_bias) v = v + bfg[i];
        if (has_gc) v = v + gcv[i][j];
        if (has_lc) v = v + lcv;
        return v;
    };
    const int n16 = lane & 15;
    auto tap0 = [&](const float* ring, unsigned pos0_, unsigned d_, unsigned roff_, unsigned t, float& xa_, float& xb_) __attribute__((always_inline)) {
        const unsigned slot = ring_slot(pos0_, t, d_);
        const unsigned long long* p = reinterpret_cast<const unsigned long long*>(ring + roff_ + slot * 32);
        const unsigned long long qa = __hip_atomic_load((gu64*)(p + (n16 >> 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long qb = __hip_atomic_load((gu64*)(p + 8 + (n16 >> 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        xa_ = __uint_as_float((n16 & 1) ? (unsigned)(qa >> 32) : (unsigned)qa);
        xb_ = __uint_as_float((n16 & 1) ? (unsigned)(qb >> 32) : (unsigned)qb);
    };
    // ---- step 0: from the persisted state
#pragma unroll
    for (int i = 0; i < kReg; ++i) {
        if (i < nown) {
            const int l = sv + 8 * i;
#pragma unroll
            for (int j = 0; j < kMS; ++j) {
                if (!ex[j]) continue;
                float xa_, xb_;
                tap0(stb[j] + L.st_ring, pos0[i][j], dil[i], roff[i], 0u, xa_, xb_);
                const float pre = dot32_dpp(t0[i].w, xa_, xb_);
                const float lcv = has_lc ? stb[j][L.st_lcprev + l * 64 + lane] : 0.0f;
                xb_store(rsv[j], (int)XcdExch::PG + l * 64, lane, 1u, addend(i, j, pre, lcv));
            }
        }
    }
    for (int t = 0; t < T && !pl.dead; ++t) {
        asm volatile(".p2align 6");                        // the step loop starts on a 64-byte line whatever was compiled in front of it (XALIGN)
        const unsigned tag = (unsigned)t + 1u;
        const bool more = t + 1 < T;
#pragma unroll
        for (int i = 0; i < kReg; ++i) {
            if (i >= nown) continue;
            const int l = sv + 8 * i;
#pragma unroll
            for (int j = 0; j < kMS; ++j) {
                if (!ex[j] || pl.dead) continue;
                const rsrc_t rs = rsv[j];
                float* ring = stb[j] + L.st_ring;
                pl.rs = rs;
                float oa = 0.0f, ob = 0.0f;
                if (more && dil[i] > 1) tap0(ring, pos0[i][j], dil[i], roff[i], (unsigned)t + 1u, oa, ob);     // already in the delay line
                unsigned long long qa, qb;
                WACC_T0();
                pl.it = 0;
                for (;;) {                                             // the layer input x_l[t] from the chain
                    qa = xbm_load(rs, (int)XcdExch::ZX + l * 128 + 1, n16 * 2);
                    qb = xbm_load(rs, (int)XcdExch::ZX + l * 128 + 65, n16 * 2);
                    if (__all(g_tag(qa) == tag && g_tag(qb) == tag)) break;
                    if (!poll_tick<true>(pl, 41)) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                WACC_T1(0);
                if (pl.dead) continue;
                const float xa_ = g_val(qa), xb_ = g_val(qb);
                const unsigned slot = ring_slot(pos0[i][j], (unsigned)t, dil[i]);      // model.py:145 dilation queue <- layer input
                if (lane < 32) ring[roff[i] + slot * 32 + lane] = lane < 16 ? xa_ : xb_;
                if (more) {
                    if (dil[i] == 1) { oa = xa_; ob = xb_; }
                    const float pre = dot32_dpp(t0[i].w, oa, ob);
                    float lcv = 0.0f;
                    if (has_lc) {                                      // frame pushed at step t is used at step t+1 (model.py:79-80)
                        const int lw = (int)XcdExch::LCR + (((t + 1) % kXcdManyLcRing) * kXcdLs + l) * 64;
                        unsigned long long ql;
                        WACC_T0();
                        pl.it = 0;
                        for (;;) {
                            ql = xbm_load(rs, lw, lane);
                            if (__all(g_tag(ql) == tag + 1u)) break;
                            if (!poll_tick<true>(pl, 42)) break;
                            __builtin_amdgcn_s_sleep(8);
                        }
                        WACC_T1(1);
                        if (pl.dead) continue;
                        lcv = g_val(ql);
                    }
                    xb_store(rs, (int)XcdExch::PG + l * 64, lane, tag + 1u, addend(i, j, pre, lcv));
                }
            }
        }
    }
    if (lane == 0) {
        if (!pl.dead) {
#pragma unroll
            for (int i = 0; i < kReg; ++i)
                if (i < nown)
#pragma unroll
                    for (int j = 0; j < kMS; ++j)
                        if (ex[j]) reinterpret_cast<int*>(stb[j] + L.st_ringpos)[sv + 8 * i] = (int)((pos0[i][j] + (unsigned)T) % dil[i]);
        } else {
#pragma unroll
            for (int j = 0; j < kMS; ++j) xb_store(rsv[j], (int)XcdExch::CTRL + 1, 0, 1u, 0.0f);
        }
    }
    WACC_OUT(a.prof, xcc == 0 ? 4 + c : -1, sv);
}

// ---- SKIP workgroup r = (layer group q, output half hh): model.py:94-96 skip 1x1 of ONE layer per wave for four 64-column slices,
//      model.py:154 sum in layer order as a relay (LDS inside the group, one L2 hop between groups), model.py:157 relu at the end ---
//      (Measured alternative, round 3: FEEDER waves that only publish their four values and one or two CLOSER waves per group that add
//      them up in order while waiting for the group's last layer -- no wave ever waits for a running total.  Bit-exact, but slower:
//      B = 8 11.3 instead of 10.5 us/step, and from six streams per XCD on the closers saturate (B = 64: 16.7 us): the adds that are
//      left when the last layer arrives sit on ONE wave's sample path instead of being spread over the relay.  Also measured: the relay
//      through every second wave (an adder takes the feeder's value before its own: one hop per two layers) -- bit-exact, 10.57 / 11.02
//      instead of 10.44 / 10.77 us/step at B = 8 / 64.)
template <int INSTR>
__device__ __forceinline__ void skip_many_role(const XArgs& xa, int xcc, int ns, int r)
{
    const XcdLaunch& a = xa.p;
    const Layout& L = a.lay;
    const int m = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int NL = L.NL, T = a.T;
    const bool use_bias = L.use_bias != 0;
    const int q = r >> 1, hh = r & 1;
    const int ngroups = (NL + 7) >> 3;
    if (q >= ngroups) return;
    const int first = NL - 8 * (ngroups - 1);                   // size of group 0 (the LAST group is always full: its relay has caught up)
    const int base = q == 0 ? 0 : first + 8 * (q - 1);
    const int size = q == 0 ? first : 8;
    if (m >= size) return;
    const int l = base + m;
    const bool from_lds = m > 0, from_l2 = (m == 0 && q > 0);
    const bool last = (l == NL - 1), to_l2 = (!last && m == size - 1);
    Tile ws[4];
    float bs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    {
        const long long lb = L.off_layer0 + (long long)l * L.layer_stride;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            load_tile(ws[j], a.P + lb + LayerOff::SK + (long long)(4 * hh + j) * kTile, lane);
            if (use_bias) bs[j] = a.P[lb + LayerOff::SK + (long long)L.NSJ * kTile + (4 * hh + j) * 64 + lane];
        }
    }
    Poll pl{exch_rsrc(a, xcc), a.status, 0, false};
    const int n16 = lane & 15;
    const int zc_lane = lane_of_z((lane < 32 ? 0 : 16) + n16);
    unsigned long long seen = 0, period = 0;
    WACC_DECL();
    for (int t = 0; t < T && !pl.dead; ++t) {
        asm volatile(".p2align 6");                        // the step loop starts on a 64-byte line whatever was compiled in front of it (XALIGN)
        const unsigned tag = (unsigned)t + 1u;
        WACC_T0();
        if (period) nap_until(seen + period - (period >> 3));
        WACC_T1(0);
        WTRACE(xcc == 0, 128 + (r * 8 + m) * 4 + 0);
#pragma nounroll
        for (int k = 0; k < ns; ++k) {
            if (pl.dead) break;
            const rsrc_t rs = exch_rsrc(a, xcc + 8 * k);
            pl.rs = rs;
            // the layer's 32 z values with ONE load per round (lanes 0-31 z[0..15] twice, lanes 32-63 z[16..31] twice)
            unsigned long long qz;
            MMARK(rs, 16 + 8 * r + m, 1);
            WACC_T0();
            pl.it = 0;
            for (;;) {
                qz = xbm_load(rs, (int)XcdExch::ZX + l * 128, zc_lane * 2);
                if (__all(g_tag(qz) == tag)) break;
                if (!poll_tick<true>(pl, 51)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            WACC_T1(1);
            MMARK(rs, 16 + 8 * r + m, 2);
            WTRACE(xcc == 0 && k == 0, 128 + (r * 8 + m) * 4 + 1);
            if (k == 0) {
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                const unsigned long long d = now - seen;
                period = (seen != 0 && d < (1ull << 18)) ? d : 0;
                seen = now;
            }
            const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)qz, (unsigned)qz, false, false);   // [lo, lo], [hi, hi]
            const float xa_ = __uint_as_float(sw[0]), xb_ = __uint_as_float(sw[1]);
            float val[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                val[j] = dot32_dpp(ws[j].w, xa_, xb_);                                              // model.py:96
                if (use_bias) val[j] = val[j] + bs[j];
            }
            // model.py:154: the running total of layers 0 .. l-1 comes from the previous wave (LDS) or the previous group (L2)
            float tot[4];
            WACC_T0();
            if (from_lds) {
                const int o = (k * 8 + m) * 64 + lane;
                f32x4 din;
                pl.it = 0;
                for (;;) {
                    const int tg = LDSVI(kManySkipData4 * 4 + o);
                    asm volatile("" ::: "memory");
                    din = LDS4(o);
                    asm volatile("" ::: "memory");
                    if (__all((unsigned)tg == tag)) break;
                    if (!poll_tick<true>(pl, 52)) break;
                }
                tot[0] = din.x + val[0]; tot[1] = din.y + val[1]; tot[2] = din.z + val[2]; tot[3] = din.w + val[3];
            } else if (from_l2) {
                u32x4s d0, d1;
                const int uw = (int)XcdExch::SKT + ((q - 1) * 2 + hh) * 256;
                pl.it = 0;
                for (;;) {
                    d0 = xbm_load2(rs, uw, lane);
                    d1 = xbm_load2(rs, uw + 128, lane);
                    if (__all(d0.y == tag && d0.w == tag && d1.y == tag && d1.w == tag)) break;
                    if (!poll_tick<true>(pl, 53)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                tot[0] = __uint_as_float(d0.x) + val[0]; tot[1] = __uint_as_float(d0.z) + val[1];
                tot[2] = __uint_as_float(d1.x) + val[2]; tot[3] = __uint_as_float(d1.z) + val[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) tot[j] = val[j];
            }
            WACC_T1(2);
            if (pl.dead) break;
            MMARK(rs, 16 + 8 * r + m, 3);
            WTRACE(xcc == 0 && k == 0, 128 + (r * 8 + m) * 4 + 3);
            if (last) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float h = tot[j] > 0.0f ? tot[j] : 0.0f;                                  // model.py:157
                    xb_store(rs, (int)XcdExch::H1 + (4 * hh + j) * 64, lane, tag, h);
                }
            } else if (to_l2) {
                const int uw = (int)XcdExch::SKT + (q * 2 + hh) * 256;
                xb_store2(rs, uw, lane, tag, tot[0], tot[1]);
                xb_store2(rs, uw + 128, lane, tag, tot[2], tot[3]);
            } else {
                const int o = (k * 8 + m + 1) * 64 + lane;
                LDS4(o) = f32x4{tot[0], tot[1], tot[2], tot[3]};
                asm volatile("" ::: "memory");
                LDSVI(kManySkipData4 * 4 + o) = (int)tag;
            }
        }
    }
    if (pl.dead && lane == 0) {
        for (int k = 0; k < ns; ++k) xb_store(exch_rsrc(a, xcc + 8 * k), (int)XcdExch::CTRL + 1, 0, 1u, 0.0f);
    }
    WACC_OUT(a.prof, xcc == 0 ? 8 + r : -1, m);
}

// ---- LC workgroups of the many-streams kernel: lc_role's arithmetic (model.py:102-111 create_upsample row by row, model.py:75-83
//      lc_filter|lc_gate of the wave's layer) with the two things that made it the busiest role at eight streams per XCD removed
//      (scripts/many_profile.py: 88 % busy, everything else waiting on it):
//      * a transposed-conv stage's output only changes when ITS phase digit (or an earlier one, or the mel frame) changes -- with
//        upsample_factor [5, 5, 12] stage 0 every 60 rows, stage 1 every 12, only the last stage every row.  Each stage's input row is
//        kept per stream in wave-private LDS and a stage is re-run only when its input changed: same fmas on the same values, so the
//        same bits, a third of the LDS round trips; the mel frame is re-read every 300 rows instead of every row;
//      * the throttle on the chain's progress is checked every fourth row, for all streams in ONE polling round.
//      This role is what saturates first beyond nine streams per XCD (B = 80: 11.5, B = 96: 13.7 us per step).  Measured and dropped:
//      two layers per wave (six register tiles, the busy waves dealt out one per SIMD) -- 1.5 us per stream and wave instead of 1.1,
//      saturated at seven streams (B = 64: 12.5 us).
template <int INSTR>
__device__ __forceinline__ void lc_many_role(const XArgs& xa, int xcc, int ns, int wg, const int prof_slot)
{
    const XcdLaunch& a = xa.p;
    const Layout& L = a.lay;
    const int v = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int NL = L.NL, T = a.T, NLC = L.NLC, Lc = L.L, lpw = xa.lc_lpw;
    const int gw = wg * 8 + v;
    const int lfirst = gw * lpw;
    int nown = NL - lfirst;
    nown = nown < 0 ? 0 : (nown > lpw ? lpw : nown);
    if (nown == 0) return;
    Poll pl{exch_rsrc(a, xcc), a.status, 0, false};
    const int* hdr = reinterpret_cast<const int*>(a.cond);
    const int mode = hdr[XH_MODE], rows = hdr[XH_ROWS];
    const float* payload = a.cond + XH_WORDS + (long long)a.B * NL * 64;
    constexpr int kLt = 4;
    Tile lt[kLt];
#pragma unroll
    for (int idx = 0; idx < kLt; ++idx) {
        if (idx < nown * NLC) {
            const int j = idx / NLC, c = idx - j * NLC;
            load_tile(lt[idx], a.P + L.off_lcw + (long long)(lfirst + j) * L.lcw_stride + (long long)c * kTile, lane);
        }
    }
    const int n16 = lane & 15;
    const int n_up = (mode == XLC_UPSAMPLED) ? 0 : L.n_up;          // rows handed over already upsampled: no stage
    int hop = 1;
    for (int i = 0; i < L.n_up; ++i) hop *= L.up[i];
    const long long need_rows = (mode == XLC_MEL) ? ((long long)T + hop - 1) / hop : T;
    if (rows < need_rows) {
        if (lane == 0) {
            atomicMax(a.status, 71);
            for (int k = 0; k < ns; ++k) xb_store(exch_rsrc(a, xcc + 8 * k), (int)XcdExch::CTRL + 1, 0, 1u, 0.0f);
        }
        return;
    }
    // wave-private LDS: per stream n_up + 1 rows of kRow floats: row 0 the input frame, row i the output of stage i - 1
    const int kRow = NLC * 32;
    const int o_wave = v * (kManyLds / 8);                          // an eighth of the workgroup's LDS per wave (see xcd_many_lc_fits)
    auto rowoff = [&](int k, int i) { return o_wave + (k * (n_up + 1) + i) * kRow; };
    int ph[4] = {0, 0, 0, 0}, frame = 0;
    WACC_DECL();
    for (int u = 0; u < T && !pl.dead; ++u) {
        const int t = u;
        (void)t;
        // ---- throttle, every fourth row for the next four: slot (u'+1) % ring is free once the chain has started step u' + 2 - ring
        WACC_T0();
        if ((u & 3) == 0 && u + 6 - kXcdManyLcRing > 0) {
            pl.it = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < kManyPerXcd; ++k) {
                    if (k < ns) {
                        const unsigned long long q = xbm_load(exch_rsrc(a, xcc + 8 * k), (int)XcdExch::CTRL, 0);
                        ok = ok && (int)g_tag(q) >= u + 6 - kXcdManyLcRing;
                    }
                }
                if (ok) break;
                if (!poll_tick<true>(pl, 72)) break;
                __builtin_amdgcn_s_sleep(16);
            }
            if (pl.dead) break;
        }
        WACC_T1(0);
        // ---- which stages see a new input at this row: stage j iff the phase digits after j are all zero (the counter's carry
        // reached digit j); the input frame itself iff every digit is zero
        int first = n_up > 0 ? n_up - 1 : 0;
#pragma unroll
        for (int j = 3; j >= 1; --j)
            if (first == j && ph[j] == 0) first = j - 1;
        const bool reload = (n_up == 0) || (first == 0 && ph[0] == 0);
        float k0[4] = {0.f, 0.f, 0.f, 0.f}, k1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n_up && i >= first) { k0[i] = a.P[L.off_up[i] + ph[i] * 2 + 0]; k1[i] = a.P[L.off_up[i] + ph[i] * 2 + 1]; }
        // ---- the common row (only the last stage sees a new input, the hparams shape, not the call's last row): TWO streams at a time in
        // one straight line -- stage, operand reads, the two sets of three chunk dots, publishes -- so that one stream's LDS round trips
        // and its store travel under the other's arithmetic (one stream after the other: 1.15 us each, 12 streams = the XCD's step at
        // batch 96).  Same operations per stream as the general body below, which takes every other row and an odd last stream.
        int kdone = 0;
        if (ns > 9 && !reload && n_up >= 1 && first == n_up - 1 && NLC == 3 && nown == 1 && u + 1 < T && Lc <= 96) {    // (up to nine streams the
                                                                      // role waits for the chains half of the time: one at a time, as measured)
            const int il = n_up - 1;
            const float t0_ = il == 0 ? k0[0] : (il == 1 ? k0[1] : (il == 2 ? k0[2] : k0[3]));
            const float t1_ = il == 0 ? k1[0] : (il == 1 ? k1[1] : (il == 2 ? k1[2] : k1[3]));
            const int l = lfirst;
            const int ring = ((int)XcdExch::LCR + (((u + 1) % kXcdManyLcRing) * kXcdLs + l) * 64);
#pragma nounroll
            for (; kdone + 1 < ns; kdone += 2) {
                WACC_T0();
                const int oa = rowoff(kdone, il), ob = rowoff(kdone + 1, il);          // stage input rows; the output rows follow at + kRow
                const rsrc_t rsa = exch_rsrc(a, xcc + 8 * kdone), rsb = exch_rsrc(a, xcc + 8 * (kdone + 1));
                // out[m] = K[a][0]*in[m] + K[a][1]*in[m-1]   as a two-term AC-1 chunk (model.py:102-111); m = lane and m = 64 + lane (< 96)
                const float xa0 = lds[oa + lane], xa1 = lane > 0 ? lds[oa + lane - 1] : 0.0f, xa2 = lane < 32 ? lds[oa + 64 + lane] : 0.0f, xa3 = lane < 32 ? lds[oa + 63 + lane] : 0.0f;
                const float xb0 = lds[ob + lane], xb1 = lane > 0 ? lds[ob + lane - 1] : 0.0f, xb2 = lane < 32 ? lds[ob + 64 + lane] : 0.0f, xb3 = lane < 32 ? lds[ob + 63 + lane] : 0.0f;
                const float ra0 = (fma_(t0_, xa0, 0.0f) + fma_(t1_, xa1, 0.0f)) + (0.0f + 0.0f), ra1 = (fma_(t0_, xa2, 0.0f) + fma_(t1_, xa3, 0.0f)) + (0.0f + 0.0f);
                const float rb0 = (fma_(t0_, xb0, 0.0f) + fma_(t1_, xb1, 0.0f)) + (0.0f + 0.0f), rb1 = (fma_(t0_, xb2, 0.0f) + fma_(t1_, xb3, 0.0f)) + (0.0f + 0.0f);
                lds[oa + kRow + lane] = lane < Lc ? ra0 : 0.0f;
                lds[ob + kRow + lane] = lane < Lc ? rb0 : 0.0f;
                if (lane < 32) { lds[oa + kRow + 64 + lane] = 64 + lane < Lc ? ra1 : 0.0f; lds[ob + kRow + 64 + lane] = 64 + lane < Lc ? rb1 : 0.0f; }
                WACC_T1(1);
                WACC_T0();
                const int ca = oa + kRow, cb = ob + kRow;
                const float a0 = lds[ca + n16], b0 = lds[ca + 16 + n16], a1 = lds[ca + 32 + n16], b1 = lds[ca + 48 + n16], a2 = lds[ca + 64 + n16], b2 = lds[ca + 80 + n16];
                const float c0 = lds[cb + n16], d0 = lds[cb + 16 + n16], c1 = lds[cb + 32 + n16], d1 = lds[cb + 48 + n16], c2 = lds[cb + 64 + n16], d2 = lds[cb + 80 + n16];
                float pa0, pa1, pa2, pb0, pb1, pb2;
                dot32_dpp_x3(lt[0].w, a0, b0, lt[1].w, a1, b1, lt[2].w, a2, b2, pa0, pa1, pa2);
                dot32_dpp_x3(lt[0].w, c0, d0, lt[1].w, c1, d1, lt[2].w, c2, d2, pb0, pb1, pb2);
                WACC_T1(2);
                WACC_T0();
                xb_store(rsa, ring, lane, (unsigned)u + 2u, (pa0 + pa1) + pa2);
                xb_store(rsb, ring, lane, (unsigned)u + 2u, (pb0 + pb1) + pb2);
                WACC_T1(3);
            }
        }
#pragma nounroll
        for (int k = kdone; k < ns; ++k) {
            const int b = xcc + 8 * k;
            const rsrc_t rs = exch_rsrc(a, b);
            WACC_T0();
            if (reload) {                                               // row `frame` (mel) / row u (upsampled), zero padded to kRow
                const float* src = payload + ((long long)b * rows + (mode == XLC_UPSAMPLED ? u : frame)) * Lc;
                const float ra = lane < Lc ? src[lane] : 0.0f;
                const float rb = 64 + lane < Lc ? src[64 + lane] : 0.0f;
                const int o = rowoff(k, 0);
                if (lane < kRow) lds[o + lane] = ra;
                if (64 + lane < kRow) lds[o + 64 + lane] = rb;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < n_up && i >= first) {
                    // out[m] = K[a][0]*in[m] + K[a][1]*in[m-1]   as a two-term AC-1 chunk   (model.py:102-111, 'same' transposed conv)
                    const int o_in = rowoff(k, i), o_out = rowoff(k, i + 1);
#pragma unroll
                    for (int hlf = 0; hlf < 2; ++hlf) {
                        const int m = hlf * 64 + lane;
                        if (m < kRow) {
                            const float x0 = lds[o_in + m];
                            const float x1 = m > 0 ? lds[o_in + m - 1] : 0.0f;
                            const float s0 = fma_(k0[i], x0, 0.0f);
                            const float s1 = fma_(k1[i], x1, 0.0f);
                            const float r = (s0 + s1) + (0.0f + 0.0f);
                            lds[o_out + m] = m < Lc ? r : 0.0f;
                        }
                    }
                }
            }
            const int o_cur = rowoff(k, n_up);
            WACC_T1(1);                                             // (row + stages)
            WACC_T0();
            // ---- projections: AC-1 chunks of 32 over the lc channels, chunk values added in order
            float resx[kLt];
#pragma unroll
            for (int idx = 0; idx < kLt; ++idx) resx[idx] = 0.0f;
            if (NLC == 3 && nown == 1) {
                // the hparams shape (80 mel channels = three chunks, one layer per wave): the three chunk dots interleaved, all six
                // operand words requested first (one after the other each dot waited for its own LDS reads: ~400 cycles per dot)
                const float a0 = lds[o_cur + n16], b0 = lds[o_cur + 16 + n16], a1 = lds[o_cur + 32 + n16], b1 = lds[o_cur + 48 + n16];
                const float a2 = lds[o_cur + 64 + n16], b2 = lds[o_cur + 80 + n16];
                dot32_dpp_x3(lt[0].w, a0, b0, lt[1].w, a1, b1, lt[2].w, a2, b2, resx[0], resx[1], resx[2]);
            } else {
#pragma unroll
                for (int idx = 0; idx < kLt; ++idx) {
                    if (idx < nown * NLC) {
                        const int c = idx % NLC;
                        const float xa_ = lds[o_cur + c * 32 + n16], xb_ = lds[o_cur + c * 32 + 16 + n16];
                        resx[idx] = dot32_dpp(lt[idx].w, xa_, xb_);
                    }
                }
            }
            WACC_T1(2);                                             // (dots)
            WACC_T0();
#pragma unroll
            for (int j = 0; j < kLt; ++j) {
                if (j < nown) {
                    float r = 0.0f;
#pragma unroll
                    for (int idx = 0; idx < kLt; ++idx) {
                        if (idx >= j * NLC && idx < (j + 1) * NLC) r = (idx == j * NLC) ? resx[idx] : r + resx[idx];
                    }
                    const int l = lfirst + j;
                    if (u + 1 < T) xb_store(rs, (int)XcdExch::LCR + (((u + 1) % kXcdManyLcRing) * kXcdLs + l) * 64, lane, (unsigned)u + 2u, r);
                    else {
                        // the frame the NEXT call uses at its step 0; in a short call wait until the service workgroup has read the
                        // previous call's frame from the same slot (its step-0 addend, PG tag 1, carries that frame; see lc_role)
                        if (T <= kXcdManyLcRing + 1) {
                            pl.rs = rs;
                            pl.it = 0;
                            for (;;) {
                                const unsigned long long ql = xbm_load(rs, (int)XcdExch::PG + l * 64, lane);
                                if (__all(g_tag(ql) >= 1u)) break;
                                if (!poll_tick<true>(pl, 73)) break;
                                __builtin_amdgcn_s_sleep(8);
                            }
                        }
                        (a.state + (long long)b * L.state_stride)[L.st_lcprev + l * 64 + lane] = r;
                    }
                }
            }
            WACC_T1(3);                                             // (combine + publish)
        }
        if (n_up > 0) {                                             // advance the phase counter (last stage fastest)
            int carry = 1;
#pragma unroll
            for (int i = 3; i >= 0; --i) {
                if (i < n_up && carry) { ph[i] += 1; carry = 0; if (ph[i] == L.up[i]) { ph[i] = 0; carry = 1; } }
            }
            frame += carry;
        }
    }
    if (pl.dead && lane == 0)
        for (int k = 0; k < ns; ++k) xb_store(exch_rsrc(a, xcc + 8 * k), (int)XcdExch::CTRL + 1, 0, 1u, 0.0f);
    WACC_OUT(a.prof, prof_slot, v);
}

template <int INSTR>
__global__ void __launch_bounds__(512) wn_xcd_many_kernel(XArgs xa)
{
    const XcdLaunch& a = xa.p;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    int ticket = 0;
    if (threadIdx.x == 0) ticket = (xcc < (unsigned)a.B && xcc < 8u) ? atomicAdd(a.roles + xcc, 1) : 1 << 20;
    __shared__ int s_ticket;
    if (threadIdx.x == 0) s_ticket = ticket;
    for (int i = threadIdx.x; i < kManyLds; i += blockDim.x) lds[i] = 0.0f;      // LDS hand-off words start at zero
    __syncthreads();
    ticket = s_ticket;
    const int ns = xcc < (unsigned)a.B && xcc < 8u ? (a.B - (int)xcc + 7) / 8 : 0;     // streams on this XCD (<= kManyPerXcd = 12)
    const int nch = (ns + kMS - 1) / kMS;                         // chain (and service) workgroups: two streams each
    if (ticket >= 2 * nch + 16 + xa.n_lc_wg) return;            // surplus workgroup (or an XCD without a stream)
    if (!roles_resident(xa)) return;                            // the device is busy: nobody starts (status 90)
    const bool forced = a.forced != nullptr;
    if (ticket < nch) {
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int nlw = a.lay.NL - (wv < 6 ? 4 * wv : 24 + 3 * (wv - 6));
        // (AC-1b: bias / gc / lc reach the chain inside the service workgroup's addend, so the chain code no longer depends on which of
        // them a model has: one set of instantiations serves every model)
        if (forced) chain_many_role<INSTR, true, true, -1>(xa, (int)xcc, ns, ticket);
        else if (nlw >= 4 && wv < 6) chain_many_role<INSTR, true, false, 4>(xa, (int)xcc, ns, ticket);
        else if (nlw >= 3 && wv >= 6) chain_many_role<INSTR, true, false, 3>(xa, (int)xcc, ns, ticket);
        else chain_many_role<INSTR, true, false, -1>(xa, (int)xcc, ns, ticket);
        return;
    }
    if (ticket < 2 * nch) { service_many_role<INSTR>(xa, (int)xcc, ns, ticket - nch); return; }
    const int role = ticket - 2 * nch;                           // 0-7 skip, 8-15 conv1, 16.. lc
    if (role < 8) {
        if (!forced) {
            skip_many_role<INSTR>(xa, (int)xcc, ns, role);
        }
        return;
    }
    if (role < 16) {
        if (!forced) {
            const XStreamsLazy<kManyPerXcd> sx{{&a, (int)xcc, ns}, {(int)xcc, ns}};
            conv1_role<(INSTR & 4), kManyPerXcd, true, XStreamsLazy<kManyPerXcd>>(xa, sx, role - 8, ns, xcc == 0 ? 16 + role - 8 : -1);
        }
    }
    else lc_many_role<INSTR>(xa, (int)xcc, ns, role - 16, xcc == 0 ? 24 + role - 16 : -1);
}

// ---- pack: the chain's register images from the canonical blob (generate.py:157-161 Saver.restore) -------------------------
__global__ void wn_xcd_pack_kernel(float* dst, const float* blob, Layout L)
{
    const long long per_layer = kXcdXlFloats;
    const long long total = per_layer * L.NL + kXcdXcFloats;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (i < per_layer * L.NL) {
            const int l = (int)(i / per_layer);
            const int e = (int)(i - (long long)l * per_layer);
            const int q = e >> 8, lane = (e >> 2) & 63, c = e & 3;          // [q][lane][4]
            const long long lb = L.c_layer0 + (long long)l * L.c_layer_stride;
            const int oc = dpp_conv_out(lane), od = dpp_dense_out(lane);
            if (q < 8) {                                                    // conv_filter|conv_gate kernel (2,R,D), tap 1
                const int k = 4 * q + c;
                v = blob[lb + (oc < 32 ? L.c_wf : L.c_wg) + 32 * 32 + k * 32 + (oc & 31)];
            } else if (q < 12) {                                            // dense kernel (1,D,R): this lane's half of the chunk
                const int i16 = 4 * (q - 8) + c;
                v = blob[lb + L.c_wd + dpp_dense_k(lane, i16) * 32 + od];
            } else if (L.use_bias) {
                if (c == 0) v = blob[lb + (oc < 32 ? L.c_bf : L.c_bg) + (oc & 31)];
                else if (c == 1) v = blob[lb + L.c_bd + od];
            }
            dst[L.off_xl + i] = v;
        } else {
            const int e = (int)(i - per_layer * L.NL);
            const int q = e >> 8, lane = (e >> 2) & 63, c = e & 3;
            const int k = 4 * q + c;                                        // wavenet/conv1d/kernel (ifw,1,R)
            if (k < L.ifw) v = blob[L.c_causal + k * 32 + dpp_dense_out(lane)];
            dst[L.off_xc + e] = v;
        }
    }
}

}  // namespace

namespace twv {

bool xcd_model_ok(const Layout& L)
{
    // scalar input: the MoL head (initial_filter_width 32, out_channels <= 32); one-hot input: the mu-law-256 model (256 classes over
    // the 8 conv1 workgroups, 32 each)
    const bool head_ok = L.scalar ? (L.ifw == 32 && L.O <= 32 && L.NOJ == 1) : (L.Q == 256 && L.O == 256);
    return head_ok && L.S == 512 && L.NL >= 1 && L.NL <= kXcdMaxLayers && L.NLC <= 4;
}
// many-streams kernel: every lc wave caches (n_up + 1) rows of NLC*32 floats for each of the XCD's eight streams in its eighth of the LDS
static bool xcd_many_lc_fits(const Layout& L) { return (long long)kManyPerXcd * (L.n_up + 1) * L.NLC * 32 <= kManyLds / 8; }
int xcd_max_streams(const Layout& L) { return L.NL > kXcdSeg0Layers ? kXcdStreams / 2 : (L.scalar && xcd_many_lc_fits(L) ? kXcdManyStreams : kXcdStreams); }
static int lc_layers_per_wave(const Layout& L) { const int n = L.NLC > 0 ? (L.NL > kXcdSeg0Layers ? 6 : 4) / L.NLC : 1; return n < 1 ? 1 : (n > 4 ? 4 : n); }
int xcd_lc_workgroups(const Layout& L)
{
    if (L.L == 0) return 0;
    const int lpw = lc_layers_per_wave(L);
    const int waves = (L.NL + lpw - 1) / lpw;
    return (waves + 7) / 8;
}
int xcd_workgroups_per_stream(const Layout& L) { return ROLE_LC0 + xcd_lc_workgroups(L); }
size_t xcd_exchange_bytes(int batch) { return (size_t)batch * XcdExch::WORDS * 8 + 64; }
void xcd_pack(float* packed, const float* blob, const Layout& L, hipStream_t st)
{
    hipLaunchKernelGGL(wn_xcd_pack_kernel, dim3(512), dim3(256), 0, st, packed, blob, L);
}
// the choice between the two XCD kernels (also what twv_wavenet_kernel_name reports): `many_opt` = option "xcd_many"
bool xcd_uses_many(const Layout& L, int batch, int many_opt)
{
    return L.scalar && (batch > kXcdStreams || many_opt == 1 || (many_opt == 0 && batch >= kXcdManyFrom)) && L.NL <= kXcdSeg0Layers && xcd_many_lc_fits(L);
}
int xcd_launch(const XcdLaunch& p, hipStream_t st)
{
    XArgs xa;
    xa.p = p;
    xa.n_lc_wg = xcd_lc_workgroups(p.lay);
    xa.lc_lpw = lc_layers_per_wave(p.lay);
    // chain workgroup: hand-off boxes + the dense kernels of its layers (136 KiB); more than 32 layers: the skip / service workgroups
    // keep the tiles of layers 0 .. NL-41 in LDS next to their value slots (159 KiB of the CU's 160)
    const bool many = xcd_uses_many(p.lay, p.B, p.many);
    xa.total_roles = 0;
    for (int x = 0; x < 8 && x < p.B; ++x) {
        const int ns = (p.B - x + 7) / 8;
        if (many) xa.total_roles += 2 * ((ns + kMS - 1) / kMS) + 16 + xa.n_lc_wg;
        else xa.total_roles += ns * (p.lay.NL > kXcdSeg0Layers ? 2 : 1) + ns + 16 + xa.n_lc_wg;
    }
    const size_t shm = many ? (size_t)kManyLds * 4 : p.lay.NL > kXcdSeg0Layers ? (size_t)159 * 1024 : (size_t)(2048 + 32 * 1024) * 4;
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    else (void)hipGetLastError();
    // twice the CUs: a workgroup takes a whole CU (512 threads x ~250 VGPRs); surplus workgroups leave at once, so every XCD's role
    // table fills whatever order the dispatcher uses
    const int grid = 2 * cus;
    // instrumented builds are separate instantiations: 1 = phase stamps / stage markers, 2 = per-layer dumps
    const int instr = (p.prof != nullptr ? 1 : 0) | (p.dbg != nullptr ? 2 : 0);
    auto go = [&](auto kern) {
        // more than 64 KiB of dynamic LDS: say so (the runtime has accepted the launch without it; the attribute is the documented way)
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        (void)hipGetLastError();
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shm, st, xa);
    };
    if (!many && instr != 0 && p.lay.NL <= TWV_XCD_KF)
        return twv_fail(TWV_E_UNSUPPORTED, "layer dumps / phase stamps of the XCD kernel need a model of at least two layers (set option \"xcd\" = 0 for the generic kernel)");
    if (many) {
        if (p.B > kXcdManyStreams) return twv_fail(TWV_E_UNSUPPORTED, "the many-streams XCD kernel takes at most 96 streams");
        // a profile buffer selects the wait-accounting build here (scripts/many_profile.py), not the phase stamps of the batch <= 32 kernel
        if (instr == 3) return twv_fail(TWV_E_UNSUPPORTED, "layer dumps and wait accounting are separate builds of the many-streams kernel");
        if (instr == 1) go(wn_xcd_many_kernel<4>);
        else if (instr == 2) go(wn_xcd_many_kernel<2>);
        else go(wn_xcd_many_kernel<0>);
    }
    else if (!p.lay.scalar) {
        // the one-hot mu-law-256 model: its own instantiations (per-layer dumps for the 30-layer kernel only; no phase stamps)
        if (p.B > (p.lay.NL > kXcdSeg0Layers ? kXcdStreams / 2 : kXcdStreams)) return twv_fail(TWV_E_UNSUPPORTED, "the one-hot XCD kernel takes at most 32 streams (16 above 30 layers)");
        if (instr == 3 || (instr != 0 && p.lay.NL > kXcdSeg0Layers))
            return twv_fail(TWV_E_UNSUPPORTED, "the one-hot XCD kernel has layer dumps OR phase stamps, up to 30 layers (set option \"xcd\" = 0 for the generic kernel)");
        if (p.lay.NL > kXcdSeg0Layers) go(wn_xcd_generate_kernel<0, true, false, true>);
        else if (instr == 2) go(wn_xcd_generate_kernel<2, false, false, true>);
        else if (instr == 1) go(wn_xcd_generate_kernel<1, false, false, true>);
        else if (p.B <= 8) go(wn_xcd_generate_kernel<0, false, true, true>);
        else go(wn_xcd_generate_kernel<0, false, false, true>);
    }
    else if (p.lay.NL > kXcdSeg0Layers) {
        if (instr != 0) return twv_fail(TWV_E_UNSUPPORTED, "layer dumps / phase stamps exist for the 30-layer XCD kernel only (set option \"xcd\" = 0 for the generic kernel)");
        go(wn_xcd_generate_kernel<0, true, false>);
    }
    else if (instr == 3) go(wn_xcd_generate_kernel<3, false, false>);
    else if (instr == 2) go(wn_xcd_generate_kernel<2, false, false>);
    else if (instr == 1) go(wn_xcd_generate_kernel<1, false, false>);
    // ONE: batch <= 8 (one stream per XCD) without the multi-stream roles compiled into the kernel: the register allocation of the
    // sampling loop depends on what shares the kernel (measured on one box, interleaved: 10.42 against 10.59 us/step; specialising
    // further -- only the hparams-default sampling chain in the kernel -- gave 10.61 again: it is allocation luck, not a trend)
    else if (p.B <= 8) go(wn_xcd_generate_kernel<0, false, true>);
    else go(wn_xcd_generate_kernel<0, false, false>);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return twv_fail(TWV_E_HIP, std::string("xcd launch: ") + hipGetErrorString(e));
    return TWV_OK;
}

}  // namespace twv
