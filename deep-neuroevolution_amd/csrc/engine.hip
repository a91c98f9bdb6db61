// engine.hip -- host side of libdne_hip.so: the C ABI of include/dne_hip.h over the gfx950 kernels in
// env_synth.h / forward.h / reduce.h.  One handle = one HIP device (a main stream + 3 window streams); all state (noise table,
// parent vectors, optimizer moments, frame stacks, emulator RAM, activations) lives in HBM for the
// lifetime of the handle, and a generation only moves (noise_idx, seed) in and (return, length) out.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only: librccl.so is opened on demand by dne_comm_init
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <array>
#include <cstdlib>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/dne_hip.h"
#include "env_synth.h"
#include "forward.h"
#include "forward_variants.h"
#include "forward_large.h"
#include "reduce.h"

using namespace dne;

static thread_local std::string g_create_error;

#define HCHECK(h, expr)                                                                         \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) return (h)->fail("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
    } while (0)

// ------------------------------------------------------------------------------- env kernels
struct EnvArgs {
    uint8_t *ram_prev, *ram_cur;     // [M][128]
    uint8_t *stacks;                 // [M][84][84][4]
    const ResizeLds *T;
    float *ret, *sign, *step_reward; // [M]
    int32_t *len, *done, *stepped;
    const int32_t *action;
    int32_t *step_counter;           // one counter per launch set: members actually stepped (profiling)
    uint8_t *bc;
    int bc_mode;                     // 0 none, 1 RAM per step (ES, policies.py:410,418), 2 final RAM (GA, policies.py:510)
    int bc_max_steps;
    int immortal;                    // DNE_DEBUG_IMMORTAL (timing experiments only): game over does not end the episode, only tslimit does
    // speculative tail (k_env_spec / k_env_render_spec / k_tail_select): the outcome of EVERY action from the current state,
    // indexed by (position in the active list, action)
    uint8_t *spec_prev, *spec_cur;   // [SPEC_CAP * 32][128] RAM rows before / after the step's last frame
    int32_t *spec_rw;                // [SPEC_CAP * 32][2]: reward, game over
    uint8_t *spec_stacks;            // [SPEC_CAP * 32][84][84][4]
    float *spec_y1;                  // [SPEC_CAP * 32][441][16]: conv1 of every candidate stack (the next step starts at conv2)
    // tail table (forward.h TailTable): position in the window -> member, in the kernel arguments; tt_n = 0: read the list
    int tt_n;
    int tt_member[TT_MAX];
};

__device__ __forceinline__ int env_member(const EnvArgs &E, const int *__restrict__ list, int gsize, int b) {
    if (E.tt_n > 0) return E.tt_member[b];
    const int g = list ? list[b / gsize] : b / gsize;
    return g * gsize + b % gsize;
}

// The emulator state is 40 live bytes per member (RAM bytes 40..127 stay zero).  The per-frame logic is branchy
// scalar code: it runs one member per lane on a register-resident copy of the state (struct Emu); the pixel work
// runs with one or more workgroups per member.
__device__ __forceinline__ Emu ram_load(const uint8_t *row) {
    uint32_t w[RAM_LIVE / 4];
#pragma unroll
    for (int j = 0; j < RAM_LIVE / 4; j++) w[j] = ((const uint32_t *)row)[j];
    return emu_unpack(w);
}
__device__ __forceinline__ void ram_store(const Emu &e, uint8_t *row) {
    uint32_t w[RAM_LIVE / 4];
    emu_pack(e, w);
#pragma unroll
    for (int j = 0; j < RAM_LIVE / 4; j++) ((uint32_t *)row)[j] = w[j];
}

// MaxAndSkipEnv (atari_wrappers.py:88-107): 4 raw frames, rewards summed, early stop on game over;
// prev / cur are the RAM before / after the last executed frame.
#ifdef DNE_PHASE_CLOCK
// Profiling build only (DNE_ENV_BURN): what the headline would do if a raw frame cost what a 6507 + TIA
// interpreter costs -- every lane that steps an emulator spends this many extra lane-instructions per raw frame on a dependent integer
// chain (four per iteration) whose result nobody reads.  The fixture's own frame is ~60; results are unchanged.
__device__ int g_env_burn = 0;
#endif
__device__ inline int skip4(Emu &prev, Emu &cur, int action, int *over) {
    int tot = 0;
    *over = 0;
    for (int i = 0; i < 4; i++) {
        prev = cur;
#ifdef DNE_PHASE_CLOCK
        {
            unsigned x = cur.rng + (unsigned)i;
            for (int b = g_env_burn >> 2; b > 0; b--) {
                x = x * 1664525u + 1013904223u;
                asm volatile("" : "+v"(x));
            }
        }
#endif
        tot += emu_frame(cur, action);
        if (cur.over) { *over = 1; break; }
    }
    return tot;
}

__global__ __launch_bounds__(64) void k_env_reset_logic(EnvArgs E, const uint32_t *__restrict__ seeds, int n) {
    const int m = blockIdx.x * 64 + threadIdx.x;
    if (m >= n) return;
    const uint32_t seed = seeds[m];
    Emu cur = emu_reset(seed), prev = cur;            // env.reset()
    const int noops = 1 + (int)(seed % 30u);         // atari_wrappers.py:18-31 (count fixed by the seed)
    for (int i = 0; i < noops; i++) { prev = cur; emu_frame(cur, 0); }
    int over;
    skip4(prev, cur, 1, &over);                      // atari_wrappers.py:40-48 FIRE then action 2
    skip4(prev, cur, 2, &over);
    uint32_t *gp = (uint32_t *)(E.ram_prev + (size_t)m * 128), *gc = (uint32_t *)(E.ram_cur + (size_t)m * 128);
    for (int j = RAM_LIVE / 4; j < 32; j++) { gp[j] = 0u; gc[j] = 0u; }
    ram_store(prev, (uint8_t *)gp);
    ram_store(cur, (uint8_t *)gc);
    E.ret[m] = 0.0f; E.sign[m] = 0.0f; E.step_reward[m] = 0.0f; E.len[m] = 0; E.done[m] = 0; E.stepped[m] = 1;
}

// the episode bookkeeping of one wrapped step of member m (policies.py:399-425) from its outcome: packed RAM before / after the
// step's last frame, reward, game over.  lds_prev / lds_cur (optional): LDS copies of the RAM rows for a renderer in the same kernel.
struct EnvBook { int t; float ret, sign; };   // a member's step count and return accumulators, read ahead of the commit
__device__ __forceinline__ void env_commit(const EnvArgs &E, int m, const uint32_t (&wp)[RAM_LIVE / 4], const uint32_t (&wc)[RAM_LIVE / 4],
                                           int r, int over, int tslimit, uint8_t *lds_prev = nullptr, uint8_t *lds_cur = nullptr,
                                           const EnvBook *book = nullptr) {
    uint8_t *gp = E.ram_prev + (size_t)m * 128, *gc = E.ram_cur + (size_t)m * 128;
#pragma unroll
    for (int j = 0; j < RAM_LIVE / 4; j++) { ((uint32_t *)gp)[j] = wp[j]; ((uint32_t *)gc)[j] = wc[j]; }
    if (lds_prev) {
#pragma unroll
        for (int j = 0; j < RAM_LIVE / 4; j++) { ((uint32_t *)lds_prev)[j] = wp[j]; ((uint32_t *)lds_cur)[j] = wc[j]; }
    }
    const int t = book ? book->t : E.len[m];
    if (E.bc_mode == 1 && t < E.bc_max_steps) {     // policies.py:410,418 RAM after every step
        uint32_t *d = (uint32_t *)(E.bc + ((size_t)m * E.bc_max_steps + t) * 128);
#pragma unroll
        for (int j = 0; j < RAM_LIVE / 4; j++) d[j] = wc[j];
    }
    if (E.bc_mode == 2) {                           // policies.py:510 final RAM
        uint32_t *d = (uint32_t *)(E.bc + (size_t)m * 128);
#pragma unroll
        for (int j = 0; j < RAM_LIVE / 4; j++) d[j] = wc[j];
    }
    E.ret[m] = (book ? book->ret : E.ret[m]) + (float)r;                               // es.py:425 rews.sum()
    E.sign[m] = (book ? book->sign : E.sign[m]) + (float)((r > 0) - (r < 0));          // es.py:423 np.sign(rews).sum()
    E.step_reward[m] = (float)r;
    E.len[m] = t + 1;
    E.stepped[m] = 1;
    if (E.step_counter) atomicAdd(E.step_counter, 1);
    if ((over && !E.immortal) || t + 1 >= tslimit) E.done[m] = 1;   // policies.py:401,424-425
}

// one wrapped step of member m (atari_wrappers.py:88-107 skip-4) + its bookkeeping; the caller is a single lane
__device__ __forceinline__ void env_member_step(const EnvArgs &E, int m, int action, int tslimit, uint8_t *lds_prev = nullptr,
                                                uint8_t *lds_cur = nullptr) {
    Emu cur = ram_load(E.ram_cur + (size_t)m * 128), prev = cur;   // skip4 overwrites prev before its first use
    int over;
    const int r = skip4(prev, cur, action, &over);
    uint32_t wp[RAM_LIVE / 4], wc[RAM_LIVE / 4];
    emu_pack(prev, wp);
    emu_pack(cur, wc);
    env_commit(E, m, wp, wc, r, over, tslimit, lds_prev, lds_cur);
}

// Speculative tail.  With a handful of members left a lock-step is a latency chain, forward pass -> emulator -> renderer, on a
// nearly idle chip.  The emulator and the renderer do not need the forward pass, only its 1-of-nact answer: while the forward
// pass of step t runs, these two kernels work out the step for EVERY action from the state after step t - 1 (one lane per
// (member, action); a candidate frame stack per action), and k_tail_select only has to adopt the candidate of the action the
// policy picks.  Nothing is predicted and nothing is ever rolled back; the arithmetic per (member, action) is env_member_step's.
constexpr int SPEC_ACTIONS = 32;   // stride of the candidate arrays (>= n_actions)
__global__ __launch_bounds__(64) void k_env_logic(EnvArgs E, const int *__restrict__ list, int gsize, int n_items, int tslimit) {
    DNE_WG_BEGIN;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b < n_items) {
        const int m = env_member(E, list, gsize, b);
        if (E.done[m]) E.stepped[m] = 0;
        else env_member_step(E, m, E.action[m], tslimit);
    }
    DNE_WG_END(3);
}

// max over the last two raw frames + WarpFrame + FrameStack for every member stepped by the logic kernel
__device__ __forceinline__ void render_body(EnvLds &s, const EnvArgs &E, int m, bool fill, int band, int nbands) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        s.ram_prev[tid] = E.ram_prev[(size_t)m * 128 + tid];
        s.ram_cur[tid] = E.ram_cur[(size_t)m * 128 + tid];
    }
    __syncthreads();
    synth_observe<1>(s, (uint32_t *)(E.stacks + (size_t)m * OB_BYTES), fill, band, nbands);
}

__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_env_render(EnvArgs E, const int *__restrict__ list, int gsize, int fill, int nbands) {
    __shared__ __attribute__((aligned(16))) EnvLds s;
    const int b = blockIdx.x / nbands, band = blockIdx.x % nbands;
    DNE_PHASE(0, 0);
    DNE_WG_BEGIN;
    const int m = env_member(E, list, gsize, b);
    const int stepped = E.stepped[m];   // asked for first, looked at once the tables are on their way
    synth_load_tables(s, E.T);
    if (!stepped) return;
    render_body(s, E, m, fill != 0, band, nbands);
    DNE_PHASE(0, 4);
    DNE_WG_END(4);
}

// Tail of a generation (a few dozen members left, every kernel boundary is a visible bubble): one workgroup per
// member finishes the policy's forward pass from the fc partial sums (k_out's arithmetic: ((s0+s1)+(s2+s3)) + bias,
// bn3, relu, the 256 x nact output layer as ordered fmaf chains, first-maximum argmax), steps the emulator with the
// chosen action on one lane, and renders the new observation -- k_out + k_env_logic + k_env_render in one launch.
struct RamLds { uint8_t ram_prev[128], ram_cur[128]; };

template <bool RENDER>
struct HeadLds {
    __attribute__((aligned(16))) std::conditional_t<RENDER, EnvLds, RamLds> s;
    float red[4][1][OUT_NA];
    float lg[32];
};

// The policy head + emulator step of one member, by one workgroup (>= 256 threads; the first 256 are the output layer's 256
// inputs: forward.h out_products / out_wave_sums).  Everything that does not depend on the fc partial sums (this thread's
// output-layer weights, the RAM rows, the resize tables) is issued before wait() -- a hook a producer / consumer variant would
// block in; the kernels in use pass NoWait.
template <bool HAS_BN, bool RENDER, bool TT, typename WaitFn>
__device__ __forceinline__ void head_body(HeadLds<RENDER> &H, const FwdArgs &A, const EnvArgs &E, int m, int tslimit,
                                          const float *__restrict__ y3t, float *__restrict__ y3, int32_t *__restrict__ actions,
                                          WaitFn wait, int pos /* position in the window (tail table), or -1 */,
                                          int spec_pos = -1 /* >= 0: adopt the speculated outcome at this list position */) {
    auto &s = H.s;
    float (&lg)[32] = H.lg;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const Layout &L = A.L;
    const int nact = L.nact;
    Item who;
    who.member = m; who.pos = TT ? pos : -1;
    const float sc = item_scale<TT>(A, who);
    const float *base = item_base<TT>(A, who);
    const float *noise_slice = item_eps<TT>(A, who);
    // speculative tail: thread a reads candidate a's outcome (and the member's bookkeeping) now, long before the choice is known
    uint32_t cwp[RAM_LIVE / 4] = {}, cwc[RAM_LIVE / 4] = {};
    int c_r = 0, c_over = 0;
    EnvBook book = {0, 0.0f, 0.0f};
    if (spec_pos >= 0 && tid < nact) {
        const size_t c = (size_t)spec_pos * SPEC_ACTIONS + tid;
#pragma unroll
        for (int j = 0; j < RAM_LIVE / 4; j++) {
            cwp[j] = ((const uint32_t *)(E.spec_prev + c * 128))[j];
            cwc[j] = ((const uint32_t *)(E.spec_cur + c * 128))[j];
        }
        c_r = E.spec_rw[2 * c]; c_over = E.spec_rw[2 * c + 1];
        book.t = E.len[m]; book.ret = E.ret[m]; book.sign = E.sign[m];
    }
    if (spec_pos < 0 && tid < 64) {
        s.ram_prev[tid] = E.ram_prev[(size_t)m * 128 + tid];
        s.ram_cur[tid] = E.ram_cur[(size_t)m * 128 + tid];
    }
    // a workgroup with more than the output layer's four waves steps the emulator for EVERY action on the lanes of its fifth wave
    // while the others wait for their weights: the emulator does not need the forward pass, only its 1-of-nact answer (the
    // speculative tail's idea inside one launch).  The lane of the chosen action commits; nothing is predicted or rolled back.
    const bool own_candidates = spec_pos < 0 && blockDim.x > 256;
    if (own_candidates && tid >= 256 && tid < 256 + nact) {
        Emu cur = ram_load(E.ram_cur + (size_t)m * 128), prev = cur;   // skip4 overwrites prev before its first use
        c_r = skip4(prev, cur, tid - 256, &c_over);
        emu_pack(prev, cwp);
        emu_pack(cur, cwc);
        book.t = E.len[m]; book.ret = E.ret[m]; book.sign = E.sign[m];
    }
    // this thread's row of the output layer (k = tid): nact consecutive weights of the base vector and of the noise slice
    float wth[OUT_NA], wep[OUT_NA];
    float fbt = 0.0f, fbe = 0.0f, s3 = 1.0f, h3 = 0.0f;
    if (tid < 256) {
        const float *wb = base + L.ow + tid * nact, *we = noise_slice + L.ow + tid * nact;
#pragma unroll
        for (int a = 0; a < OUT_NA; a++) {
            wth[a] = a < nact ? wb[a] : 0.0f;
            wep[a] = a < nact ? we[a] : 0.0f;
        }
        fbt = base[L.fcb + tid]; fbe = noise_slice[L.fcb + tid];
        if (HAS_BN) { s3 = A.bn[(size_t)m * 608 + 96 + tid]; h3 = A.bn[(size_t)m * 608 + 352 + tid]; }
    }
    if constexpr (RENDER) synth_load_tables(s, E.T);
    if (!wait()) return;
    DNE_PHASE(3, 1);
    if (tid < 256) {
        float t = fc_combine(y3t, m, tid, A.sub_sums != 0);
        float pvb = sc * fbe;
        const float fb = fbt + pvb;
        t = t + fb;
        y3[(size_t)m * 256 + tid] = t;
        if (HAS_BN) {
            t = t * s3;
            t = t + h3;
        }
        const float x = t > 0.0f ? t : 0.0f;
        float pr[1][OUT_NA];
#pragma unroll
        for (int a = 0; a < OUT_NA; a++) {
            float pv = sc * wep[a];
            float w = wth[a] + pv;
            pr[0][a] = x * w;
        }
        out_wave_sums<1>(pr, nact, H.red[wv], lane);
    }
    __syncthreads();
    DNE_PHASE(3, 2);
    if (tid < nact) {
        const float s01 = H.red[0][0][tid] + H.red[1][0][tid];
        const float s23 = H.red[2][0][tid] + H.red[3][0][tid];
        const float t = s01 + s23;
        float pv = sc * noise_slice[L.ob + tid];
        const float bias = base[L.ob + tid] + pv;
        lg[tid] = t + bias;
    }
    __syncthreads();
    DNE_PHASE(3, 3);
    if (spec_pos >= 0) {   // the emulator + renderer outcome of every action is on the table: every thread finds the policy's
        int best = 0;      // choice, the thread that read that candidate commits it, all copy its frame stack
        for (int a = 1; a < nact; a++)
            if (lg[a] > lg[best]) best = a;   // tf.argmax: first maximum
        if (tid == best) {
            actions[m] = best;
            env_commit(E, m, cwp, cwc, c_r, c_over, tslimit, nullptr, nullptr, &book);
        }
        const uint4 *src = (const uint4 *)(E.spec_stacks + ((size_t)spec_pos * SPEC_ACTIONS + best) * OB_BYTES);
        uint4 *dst = (uint4 *)(E.stacks + (size_t)m * OB_BYTES);
        for (int i0 = tid; i0 < OB_BYTES / 16; i0 += 8 * blockDim.x) {   // all of a thread's loads in flight before its first store
            uint4 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { const int i = i0 + j * blockDim.x; if (i < OB_BYTES / 16) v[j] = src[i]; }
#pragma unroll
            for (int j = 0; j < 8; j++) { const int i = i0 + j * blockDim.x; if (i < OB_BYTES / 16) dst[i] = v[j]; }
        }
        return;
    }
    if (own_candidates) {
        int best = 0;
        for (int a = 1; a < nact; a++)
            if (lg[a] > lg[best]) best = a;   // tf.argmax: first maximum
        if (tid == 256 + best) {
            actions[m] = best;
            env_commit(E, m, cwp, cwc, c_r, c_over, tslimit, RENDER ? s.ram_prev : nullptr, RENDER ? s.ram_cur : nullptr, &book);
        }
    } else if (tid == 0) {
        int best = 0;
        for (int a = 1; a < nact; a++)
            if (lg[a] > lg[best]) best = a;   // tf.argmax: first maximum
        actions[m] = best;
        env_member_step(E, m, best, tslimit, RENDER ? s.ram_prev : nullptr, RENDER ? s.ram_cur : nullptr);
    }
    if constexpr (RENDER) {
        __syncthreads();
        synth_observe(s, (uint32_t *)(E.stacks + (size_t)m * OB_BYTES), false);
    }
}

template <bool HAS_BN, bool RENDER>
__global__ __launch_bounds__(1024) void k_tail_step(FwdArgs A, EnvArgs E, const int *__restrict__ list, int gsize, int tslimit,
                                                     const float *__restrict__ y3t, float *__restrict__ y3,
                                                     int32_t *__restrict__ actions) {
    __shared__ HeadLds<RENDER> H;
    const int b = blockIdx.x;
    DNE_PHASE(3, 0);
    const int m = env_member(E, list, gsize, b);
    if (E.done[m]) { if (threadIdx.x == 0) E.stepped[m] = 0; return; }
    if (A.tt.n > 0) head_body<HAS_BN, RENDER, true>(H, A, E, m, tslimit, y3t, y3, actions, NoWait{}, b);
    else head_body<HAS_BN, RENDER, false>(H, A, E, m, tslimit, y3t, y3, actions, NoWait{}, b);
#ifdef DNE_PHASE_CLOCK
    __syncthreads();
    DNE_PHASE(3, 4);
#endif
}

// The two speculative kernels ride in launches of the forward pass (same stream, no event traffic -- a cross-stream event
// hand-over per lock-step costs more than the lock-step): the emulator lanes share conv1's launch, the candidate renderers share
// the quad fc's (they need the emulator's output, i.e. a kernel boundary, and the fc is the longest stage to hide under).
__global__ __launch_bounds__(256) void k_conv1_spec(FwdArgs A, EnvArgs E, const int *__restrict__ list, int gsize,
                                                    float *__restrict__ y1, int nsplit, int n_conv_blocks, int n_items, int nact) {
    __shared__ Conv1Lds S;
    if ((int)blockIdx.x < n_conv_blocks) {
        if (A.tt.n > 0) {
            const Item it = decode_item<true>(A, blockIdx.x / nsplit, list, gsize, 1, 0, E.stacks, nullptr, A.done);
            if (it.skip) return;
            conv1_body<16, true>(S, A, it, y1, blockIdx.x % nsplit, nsplit);
        } else {
            const Item it = decode_item(A, blockIdx.x / nsplit, list, gsize, 1, 0, E.stacks, nullptr, A.done);
            if (it.skip) return;
            conv1_body(S, A, it, y1, blockIdx.x % nsplit, nsplit);
        }
        return;
    }
    const int i = ((int)blockIdx.x - n_conv_blocks) * 256 + threadIdx.x;
    if (i >= n_items * nact) return;
    const int b = i / nact, a = i % nact;
    const int m = env_member(E, list, gsize, b);
    if (E.done[m]) return;
    Emu cur = ram_load(E.ram_cur + (size_t)m * 128), prev = cur;
    int over;
    const int r = skip4(prev, cur, a, &over);
    const size_t c = (size_t)b * SPEC_ACTIONS + a;
    ram_store(prev, E.spec_prev + c * 128);
    ram_store(cur, E.spec_cur + c * 128);
    E.spec_rw[2 * c] = r;
    E.spec_rw[2 * c + 1] = over;
}

// conv2 of the forward pass + the emulator outcome of every action.  y1_cand: the member's conv1 output is the candidate its
// last action selected (k_tail_select_conv1 of the previous lock-step), not a row of y1.
template <bool HAS_BN>
__global__ __launch_bounds__(256) void k_conv2_spec(FwdArgs A, EnvArgs E, const int *__restrict__ list, int gsize,
                                                    const float *__restrict__ y1, float *__restrict__ y2, int nsplit, int n_conv_blocks,
                                                    int n_items, int nact, const int32_t *__restrict__ last_action) {
    __shared__ Conv2Lds S;
    if ((int)blockIdx.x < n_conv_blocks) {
        const int b = blockIdx.x / nsplit;
        if (A.tt.n > 0) {
            const Item it = decode_item<true>(A, b, list, gsize, 1, 0, nullptr, nullptr, A.done);
            if (it.skip) return;
            const float *row = E.spec_y1 + ((size_t)b * SPEC_ACTIONS + last_action[it.member]) * 7056;
            conv2_body<HAS_BN, true>(S, A, it, y1, y2, blockIdx.x % nsplit, nsplit, nullptr, row);
        } else {
            const Item it = decode_item(A, b, list, gsize, 1, 0, nullptr, nullptr, A.done);
            if (it.skip) return;
            const float *row = E.spec_y1 + ((size_t)b * SPEC_ACTIONS + last_action[it.member]) * 7056;
            conv2_body<HAS_BN>(S, A, it, y1, y2, blockIdx.x % nsplit, nsplit, nullptr, row);
        }
        return;
    }
    const int i = ((int)blockIdx.x - n_conv_blocks) * 256 + threadIdx.x;
    if (i >= n_items * nact) return;
    const int b = i / nact, a = i % nact;
    const int m = env_member(E, list, gsize, b);
    if (E.done[m]) return;
    Emu cur = ram_load(E.ram_cur + (size_t)m * 128), prev = cur;
    int over;
    const int r = skip4(prev, cur, a, &over);
    const size_t c = (size_t)b * SPEC_ACTIONS + a;
    ram_store(prev, E.spec_prev + c * 128);
    ram_store(cur, E.spec_cur + c * 128);
    E.spec_rw[2 * c] = r;
    E.spec_rw[2 * c + 1] = over;
}

template <int NV, bool HAS_BN>
__global__ __launch_bounds__(256) void k_fc_quad_spec(FwdArgs A, EnvArgs E, const int *__restrict__ list, int gsize,
                                                      const float *__restrict__ y2, float *__restrict__ y3t, int n_fc_blocks,
                                                      int nact, int nbands) {
    constexpr size_t LDS_BYTES = sizeof(EnvLds) > sizeof(QuadLds<NV>) ? sizeof(EnvLds) : sizeof(QuadLds<NV>);
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    if ((int)blockIdx.x < n_fc_blocks) {
        QuadLds<NV> &S = *reinterpret_cast<QuadLds<NV> *>(lds);
        const int item = blockIdx.x >> 6, cg = (blockIdx.x >> 2) & 15, sl = blockIdx.x & 3;
        if (A.tt.n > 0) fc_quad_body<NV, HAS_BN, true>(S, A, list, item, cg, sl, y2, y3t);
        else fc_quad_body<NV, HAS_BN, false>(S, A, list, item, cg, sl, y2, y3t);
        return;
    }
    EnvLds &s = *reinterpret_cast<EnvLds *>(lds);
    const int rb = (int)blockIdx.x - n_fc_blocks;
    const int band = rb % nbands, ba = rb / nbands, b = ba / nact, a = ba % nact;
    const int m = env_member(E, list, gsize, b);
    if (E.done[m]) return;
    synth_load_tables(s, E.T);
    const size_t c = (size_t)b * SPEC_ACTIONS + a;
    const int tid = threadIdx.x;
    if (tid < 64) {
        s.ram_prev[tid] = E.spec_prev[c * 128 + tid];
        s.ram_cur[tid] = E.spec_cur[c * 128 + tid];
    }
    __syncthreads();
    synth_observe(s, (uint32_t *)(E.spec_stacks + c * OB_BYTES), false, band, nbands, (const uint32_t *)(E.stacks + (size_t)m * OB_BYTES));
}

template <bool HAS_BN>
__global__ __launch_bounds__(1024) void k_tail_select(FwdArgs A, EnvArgs E, const int *__restrict__ list, int gsize, int tslimit,
                                                       const float *__restrict__ y3t, float *__restrict__ y3,
                                                       int32_t *__restrict__ actions) {
    __shared__ HeadLds<false> H;
    const int b = blockIdx.x;
    const int m = env_member(E, list, gsize, b);
    if (E.done[m]) { if (threadIdx.x == 0) E.stepped[m] = 0; return; }
    if (A.tt.n > 0) head_body<HAS_BN, false, true>(H, A, E, m, tslimit, y3t, y3, actions, NoWait{}, b, b);
    else head_body<HAS_BN, false, false>(H, A, E, m, tslimit, y3t, y3, actions, NoWait{}, b, b);
}

// k_tail_select + conv1 of every candidate frame stack (workgroups past the first n_items): whichever action the policy picks in
// the same launch, the next lock-step finds its conv1 output ready and starts at conv2.
template <bool HAS_BN>
__global__ __launch_bounds__(256) void k_tail_select_conv1(FwdArgs A, EnvArgs E, const int *__restrict__ list, int gsize, int tslimit,
                                                           const float *__restrict__ y3t, float *__restrict__ y3,
                                                           int32_t *__restrict__ actions, int n_items, int nact, int nsplit) {
    constexpr size_t LDS_BYTES = sizeof(HeadLds<false>) > sizeof(Conv1Lds) ? sizeof(HeadLds<false>) : sizeof(Conv1Lds);
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    if ((int)blockIdx.x < n_items) {
        HeadLds<false> &H = *reinterpret_cast<HeadLds<false> *>(lds);
        const int b = blockIdx.x;
        const int m = env_member(E, list, gsize, b);
        if (E.done[m]) { if (threadIdx.x == 0) E.stepped[m] = 0; return; }
        if (A.tt.n > 0) head_body<HAS_BN, false, true>(H, A, E, m, tslimit, y3t, y3, actions, NoWait{}, b, b);
        else head_body<HAS_BN, false, false>(H, A, E, m, tslimit, y3t, y3, actions, NoWait{}, b, b);
        return;
    }
    Conv1Lds &S = *reinterpret_cast<Conv1Lds *>(lds);
    const int cb = (int)blockIdx.x - n_items, part = cb % nsplit, ba = cb / nsplit, b = ba / nact, a = ba % nact;
    Item it;
    it.member = env_member(E, list, gsize, b);
    it.pos = A.tt.n > 0 ? b : -1;
    if (E.done[it.member]) return;
    it.row = b * SPEC_ACTIONS + a;
    it.ob = E.spec_stacks + (size_t)it.row * OB_BYTES;
    it.skip = false;
    if (A.tt.n > 0) conv1_body<16, true>(S, A, it, E.spec_y1, part, nsplit);
    else conv1_body(S, A, it, E.spec_y1, part, nsplit);
}

// order-preserving compaction of the active-group list
__global__ __launch_bounds__(1024) void k_compact(const int32_t *__restrict__ done, int gsize, const int *__restrict__ list_in,
                                                  int n_in, int *__restrict__ list_out, int *__restrict__ count_out) {
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    int running = 0;
    for (int start = 0; start < n_in; start += 1024) {
        const int i = start + tid;
        int g = -1;
        bool alive = false;
        if (i < n_in) {
            g = list_in ? list_in[i] : i;
            for (int k = 0; k < gsize; k++) alive = alive || !done[g * gsize + k];
        }
        const unsigned long long bal = __ballot(alive);
        if (lane == 0) wave_tot[wv] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
        for (int k = 0; k < 16; k++) { before += k < wv ? wave_tot[k] : 0; total += wave_tot[k]; }
        if (alive) {
            const int pos = running + before + __popcll(bal & ((1ull << lane) - 1ull));
            list_out[pos] = g;
            if (pos < TT_MAX) count_out[8 + pos] = g;   // the head of the list travels to the host with the count (tail table)
        }
        running += total;
        __syncthreads();
    }
    if (tid == 0) *count_out = running;
}

__global__ void k_iota(int *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

// ------------------------------------------------------------------------------- handle
struct dne_handle {
    dne_config cfg{};
    Layout L{};
    std::string err;
    hipStream_t stream = nullptr;
    std::vector<hipStream_t> sub_streams;   // sub-batch streams (sub_streams[0] == stream)
    int dbg_skip = 0;   // DNE_DEBUG_SKIP bitmask (timing experiments only): 1 conv1, 2 conv2, 4 render
    int dbg_immortal = 0;   // DNE_DEBUG_IMMORTAL (timing experiments only): every member lives until tslimit -- a lock-step keeps its width
    int render_threads = 512;        // DNE_RENDER_THREADS: threads per k_env_render workgroup above 192 members (one workgroup per member).  Round 5: 512 -- beside k_fc_ring a generation takes 216.8 instead of 223.1 ms same-box (768: 217.2, 1024: 235.3); rounds 1-4: 256 (beside k_fc_duo 512 measured +1 %)
    int conv1_fpw = 8;               // reference pass: frames per conv1 workgroup (DNE_CONV1_FPW: 1, 2, 4, 8)
    int conv_fused = 1, conv_fused_min = 129;   // DNE_CONV_FUSED / DNE_CONV_FUSED_MIN: conv1 + conv2 in one kernel from this many members
    int conv12t_max = 64;            // members up to which conv1 -> conv2 is one launch of four workgroups per member (DNE_CONV12T_MAX, 0 = off)
    int conv_split_max = 32;         // members up to which the convolutions use their finest split (DNE_CONV_SPLIT_MAX)
    int conv_split_mid = 256;        // ... their middle split: conv1 over 4 workgroups up to this many members, conv2 over 2 up to twice as many (DNE_CONV_SPLIT_MID)
    int fc_pairs = 2;                // ES full-width fc: antithetic pairs per work item (DNE_FC_PAIRS, 1 = k_fc<2>)
    int fc2_min_total = 800;         // k_fc2 from this many active groups upwards (DNE_FC2_MIN)
    bool fc2_now = false;            // decided per burst by eval_core
    int duo_solo_below = 1500;       // DNE_DUO_SOLO_BELOW: with fewer active groups (all windows) every wave takes one unit instead of two (sparse table: little to share, and twice the waves)
    bool duo_solo_now = false;       // decided per burst by eval_core
    int out_lds_kb = 0;              // DNE_OUT_LDS_KB: an unused LDS reservation that bounds k_out's workgroups per CU (round 2's k_out staged 37 KB of output
                                     // weights and ran best at two workgroups per CU = 64 KB; round 3's stages nothing)
    int duo_head_fused = 0;          // DNE_DUO_HEAD_FUSED: behind k_fc_duo the policy head and the emulator step share a launch (k_tail_step) instead of
                                     // k_out + k_env_logic; same-box A/B: 402.7 ms fused, 403.3 separate, 399.8 separate with k_out at two workgroups per CU -> off
    int conv2_ref_fpw = 8;           // DNE_CONV2_REF_FPW: reference frames per conv2 workgroup (8, 4, or 1 = the lock-step kernel)
    int duo_sync = 1;                // DNE_DUO_SYNC: row blocks per barrier of the sweep (1-8)
    int burst = 32, burst_tail = 16; // DNE_BURST / DNE_BURST_TAIL: lock-steps between two compactions of the active list (a host round trip each), at large / with at most fc_tail_max groups alive (round 4: 32 at large, 16 before; 24 / 32 / 48 measured -0.5 .. -0.9 %, 8 +2.3 %, 64 +0.2 %; the tail indifferent)
    double dense_scale = 1.0;        // table length / the stretch of the table this evaluation's noise slices cover (dne_es_eval; 1 for every other caller): a rank that draws
                                     // its indices from its own 1/N of the table (es.py shard 'table') holds pairs as dense as N times as many over the whole table
    int sub_render_fused = 0;        // DNE_SUB_RENDER_FUSED (round 6): behind the sub-slice fc the policy head, the emulator AND the renderer in one launch (k_tail_step<.., true>, 1024 threads per member): three launches per window and lock-step instead of four
    int ring_min = 1000;             // DNE_RING_MIN: k_fc_ring needs this many active pairs on the rank whatever their density (below, its workgroups -- eight units, one per CU --
                                     // no longer fill the chip: a 625-pair share measured 61.6 ms per generation on the ring against 57.8 on k_fc_duo, profiles/r06_shard_ab.jsonl)
    int ring_on = 1;                 // DNE_FC_RING (round 5): k_fc_ring instead of k_fc_duo -- the workgroup's noise rows through an LDS ring filled by LDS-DMA, the base rows from a column-permuted copy of the fc matrix; 1: from DNE_DUO_SOLO_BELOW active pairs (1500) upwards, 2: in the whole k_fc_duo range (measured slower in the sparse part: 239 vs 233 ms), 0: k_fc_duo everywhere
    float *noise_pre = nullptr;      // the noise table scaled by the evaluation's sigma, fl(sigma * eps) entry by entry (k_scale_table): k_fc_ring<true>'s DMA source.  Built the first
                                     // time an evaluation enters the ring's range at that sigma (ES runs keep one sigma: once per run, 0.4 ms), dropped with the table
    size_t noise_pre_count = 0;      // entries of the table the copy was made from (0: none)
    float noise_pre_sigma = 0.0f;    // the sigma it holds
    int ring_pre = 1;                // DNE_RING_PRE: 0 = the ring multiplies every row by sigma itself (rounds 5-6a)
    bool ring_pre_now = false;       // this evaluation's ring launches read noise_pre
    bool pair_sigma_uniform = false; // every pair of the member set is (+s, -s) with ONE s (dne_set_members)
    float pair_sigma = 0.0f;
    float *theta_perm = nullptr;     // [3872 + 16][256]: base slot 0's fc matrix, every row stored as columns l, l+64, l+128, l+192 per lane (k_theta_perm, once per evaluation)
    int duo_fat = 1;                 // DNE_DUO_FAT: k_fc_duo with a register footprint past 256 per lane = at most one of its workgroups per CU (it streams as fast from one), so the other windows' kernels always find room beside it
    int fc_sub = 1;                  // DNE_FC_SUB (ES 2, GA 1): the sub-slice fc (k_fc_sub: one wave per 128 / 120-row chain) in the mid range -- 0 off, 1 GA children (materialised), 2 also ES pairs
    int fc_sub_min = 97, fc_sub_max = 320;   // DNE_FC_SUB_MIN / _MAX: active groups (all windows) between which it runs (max: 450 for ES pairs, 320 for GA children)
    int fc_sub_nsub = 2;             // DNE_FC_SUB_NSUB: windows of that regime (ES 3, GA 4 since round 6)
    int fc_sub_spw = 0;              // DNE_FC_SUB_SPW: sub-slices per wave (1, 2, 4, 8; 0 = by width)
    int fc_sub_grid = 512;           // DNE_FC_SUB_GRID: workgroups of k_fc_sub at most, 4 waves each (GA 512 = two waves per SIMD; ES: the whole launch resident)
    int fc_sub_prio = 0;             // DNE_FC_SUB_PRIO: s_setprio of k_fc_sub's waves (ES 0: the regime is bound by a window's chain of small kernels, they must not starve; GA: 0 since round 6, 3 on its bounded grid before)
    int fc_sub_head = 1;             // DNE_FC_SUB_HEAD: policy head + emulator step in one launch (k_tail_step) behind k_fc_sub instead of k_out + k_env_logic
    bool sub_now = false;            // decided per burst by eval_core
    float *y3s = nullptr;            // [member][32][256]: the chain sums k_fc_sub leaves for k_out<.., SUB>
    int duo_grid = 0;                // DNE_DUO_GRID: persistent grid of k_fc_duo (0 = fc_grid)
    int duo_sweep = 2;               // DNE_DUO_SWEEP (0 = off): the four waves of a k_fc_duo workgroup walk one table timeline (1: two units per wave only, 2: also one unit per wave)
    int fc_prio = 3;                 // DNE_FC_PRIO: s_setprio of k_fc_duo's waves (0-3)
    int duo_lag = 0;                 // DNE_DUO_LAG: extra row batches by which the second unit of a duo trails the first
    int fc_duo = 1, fc_duo_min = 451;   // DNE_FC_DUO / DNE_FC_DUO_MIN: table-ordered fc (k_unit_order + k_fc_duo + k_out) from this many active groups (round 4: 451, right above the sub-slice fc's range; 800 before -- with one k_fc_duo workgroup per CU a 625-pair share takes 84.5 instead of 88.4 ms per generation)
    bool duo_now = false;            // decided per burst by eval_core (the unit order of each window is rebuilt then)
    int *unit_order = nullptr;       // [4 * groups]: per window, its (group, k-slice) units in noise-table order
    int spec_max = 4;                // DNE_SPEC_MAX: a single window of up to this many members (2 antithetic pairs; round 2: 8 -- the faster tail kernels of round 3 beat speculation at 4 pairs, 47 vs 56 us) steps speculatively -- every action's outcome is worked out under the forward pass; 0 = off
    uint8_t *spec_prev = nullptr, *spec_cur = nullptr, *spec_stacks = nullptr;
    int32_t *spec_rw = nullptr;
    float *spec_y1 = nullptr;
    int spec_bands = 7;              // DNE_SPEC_BANDS: 256-thread workgroups per candidate frame
    int spec_conv1 = 1;              // DNE_SPEC_CONV1: conv1 of every candidate stack in the launch that picks the action
    bool ring_now = false;           // this burst's table-ordered windows run k_fc_ring (their convolutions leave activated y2)
    bool uniform_base = false;       // every member perturbs the same base slot (dne_set_members checks)
    bool antithetic_slot0 = false;   // what k_theta_perm + k_fc_ring assume, checked member by member in dne_set_members: base slot 0 everywhere and
                                     // members (2i, 2i+1) = (offset, +s), (the same offset, -s) -- es.py:412-419's pairs, nothing else
    int render_bands = 8, band_threads = 512;   // tail: workgroups per frame (DNE_RENDER_BANDS, 1 = render inside k_tail_step) and their size
    int render_wg_max = 512;         // ... halved until members x bands fits this many workgroups (DNE_RENDER_WG_MAX)
    int tail_fused_max = 200;        // up to this many active groups (all windows) k_out + k_env_logic + k_env_render run as one kernel (DNE_TAIL_FUSED_MAX)
    int nsub_mid = 4;                // DNE_NSUB_MID: windows between 800 and 1899 active groups
    int nsub_full = 4;               // DNE_NSUB_FULL: windows at full width (>= 1900 active groups)
    int nsub_fixed = 0, fc_grid = 512, fc_tail_max = 96, fc_rb = 4, fc_chain_min = 1 << 30;
    int fc_tailk_max = 32;           // DNE_FC_TAILK_MAX: up to this many groups per window k_fc_tail (16 workgroups per group), above it k_fc_cols (4 lean ones)
    int fc_quad_max = 4;             // DNE_FC_QUAD_MAX: up to this many groups per window the 64-workgroups-per-group fc (k_fc_quad); above it k_fc_tail
    int M = 0, F = 0, ref_chunk = 0;
    size_t base_stride = 0;
    // device memory
    float *noise = nullptr; size_t noise_count = 0;
    float *bases = nullptr; int base_cap = 0;
    float *opt_m = nullptr, *opt_v = nullptr, *g = nullptr; int opt_t = 0;
    double *partial = nullptr;
    uint8_t *ref = nullptr; bool ref_set = false;
    float *ref_f32 = nullptr;        // the reference frames as padded planar floats (k_conv1_ref_shared)
    int conv1_shared = 1;            // DNE_CONV1_SHARED: reference-pass conv1 with eight members sharing a frame in LDS
    int32_t *m_slot = nullptr; int64_t *m_off = nullptr; float *m_scale = nullptr;
    std::vector<int32_t> host_slot; std::vector<int64_t> host_off; std::vector<float> host_scale;   // what dne_set_members uploaded
    TailTable tt{}; bool tt_on = false;   // the current burst's window as kernel arguments (at most TT_MAX members left)
    int tt_enable = 1;               // DNE_TAIL_TABLE
    int head_threads = 320;          // DNE_HEAD_THREADS: 320 = the policy head's four waves + a fifth that steps the emulator for every action meanwhile; 256 = one lane steps it afterwards
    float *bn = nullptr, *bn_mom = nullptr;
    uint8_t *ram_prev = nullptr, *ram_cur = nullptr, *stacks = nullptr;
    ResizeLds *tables = nullptr;
    float *ret = nullptr, *sign = nullptr, *step_reward = nullptr, *logits = nullptr;
    int32_t *len = nullptr, *done = nullptr, *action = nullptr, *stepped = nullptr;
    int32_t *launch_units = nullptr; size_t launch_units_cap = 0;
    uint32_t *seeds = nullptr;
    float *y1 = nullptr, *y2 = nullptr, *y3 = nullptr, *y3t = nullptr;   // step mode: one row per member (y3t: 4 k-slice partials)
    int ga_materialize = 0;          // DNE_GA_MATERIALIZE: GA children written out once per generation, the streaming fc then reads plain rows (default: on for GA engines)
    bool members_materialized = false;   // the current members are plain vectors (scale 0 everywhere): kernels that have one skip the noise stream
    std::vector<int> child_slots;    // base slots set aside for materialised children
    int lfc_pad = 2;                 // DNE_LFC_PAD: LargeModel's streamed fc with a padded register footprint -- 1: at most two of its workgroups per CU, 2: one (0: as many as fit, five); same-box 282.5 / 285.2 / 291.2 k env-steps/s at 0 / 1 / 2
    int lfc_cols_max = 96;           // DNE_LFC_COLS_MAX: LargeModel windows of up to this many members use the column-split fc
    bool large = false;              // DNE_KIND_GA_LARGE: y1 [441][32], y2 / y3 [121][64] (conv3 output), y3t = the 512 fc outputs
    float *y1r[2] = {nullptr, nullptr}, *y2r[2] = {nullptr, nullptr}, *y3pr[2] = {nullptr, nullptr};   // reference pass scratch, two ways
    float *fr1[2] = {nullptr, nullptr}, *fr2[2] = {nullptr, nullptr};   // per-frame batch-norm moments of conv1 / conv2 ([rows][2][C])
    hipEvent_t ev_ref[2] = {nullptr, nullptr};
    int *list_a = nullptr, *list_b = nullptr, *count_dev = nullptr;
    int *count_host = nullptr;       // pinned: the active count comes back once per burst (a pageable destination makes the copy a staged, synchronous one)
    uint8_t *bc = nullptr; size_t bc_bytes = 0;
    float *mat_out = nullptr; size_t mat_cap = 0;
    float *scratch_f = nullptr; size_t scratch_cap = 0;   // small float scratch (ranks, weights)
    int64_t *scratch_i = nullptr;
    // profiling
    std::vector<hipEvent_t> ev_pool;
    std::vector<hipEvent_t> fc_ring;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    dne_profile prof{};
    // GA parent cache: prefix chain -> base slot
    std::map<std::vector<int64_t>, int> ga_cache;
    int ga_sort = 1;                 // DNE_GA_SORT: evaluate a generation's children grouped by parent
    int ga_cache_mode = 0; float ga_cache_sigma = 0.0f;   // what the cached parents were built with (1 sigma / 2 per-seed powers)
    std::vector<int> free_slots;
    // every device allocation of the handle, each between two poisoned red zones (dne_check_redzones)
    struct Block { std::string name; uint8_t *raw; size_t bytes; };
    std::vector<Block> blocks;
    size_t redzone = 4096;           // bytes on each side (DNE_REDZONE=0 disables)
    bool trace_on = false;           // DNE_TRACE=1: stage breadcrumbs on stderr
    bool staged_copies = true;       // DNE_STAGED_COPY=0: hand large host buffers straight to hipMemcpy (diagnosis only)
    bool debug_sync = false;         // DNE_DEBUG_SYNC=1: synchronize + check after every launch set of an evaluation
    // pinned staging for large host <-> device transfers (the runtime would otherwise pin the caller's pages)
    uint8_t *stage_buf[2] = {nullptr, nullptr};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    // gathered per-pair records of one generation, device-resident (dne_allgather_results / dne_records_set)
    int64_t *rec_idx = nullptr; float *rec_ret = nullptr, *rec_sign = nullptr; int32_t *rec_len = nullptr;
    uint8_t *rec_send = nullptr, *rec_recv = nullptr; size_t rec_cap = 0, rec_wire_cap = 0; int rec_n = 0;
    std::vector<int64_t> rec_idx_host;
    int64_t *chain_offs = nullptr; float *chain_pw = nullptr; size_t chain_cap = 0;   // GA: seed offsets (+ per-seed powers) of the chain being rebuilt
    float *init_scale = nullptr;     // GA genomes in the gpu tree's form: root = noise[idx0] * init_scale (dne_ga_set_init_scale)
    // novelty archive resident on the device (dne_archive_append): rows of all entries back to back
    uint8_t *arch = nullptr; size_t arch_cap = 0, arch_rows = 0; int arch_dim = 0;
    int64_t *arch_row0 = nullptr; int32_t *arch_len = nullptr; size_t arch_ent_cap = 0;
    std::vector<int64_t> arch_row0_host; std::vector<int32_t> arch_len_host;
    long long *nov_out = nullptr; size_t nov_out_cap = 0; int32_t *nov_len = nullptr; size_t nov_len_cap = 0;   // GA: the seed offsets of the chain being rebuilt
    // RCCL communicator (dne_comm_init); the library is opened on demand
    void *rccl_lib = nullptr; void *comm = nullptr; int comm_rank = 0, comm_size = 1;
    bool comm_borrowed = false;      // dne_comm_share: the communicator belongs to another handle of this process
    std::mutex comm_mu;                  // dne_comm_init publishes its communicator and dne_comm_abort retires one under this lock
    std::atomic<bool> comm_off{false};   // dne_comm_abort (possibly from another host thread than a dne_comm_init still in flight): this handle takes no part in RCCL any more, a late result is dropped
    double *comm_scratch = nullptr;

    template <typename T>
    hipError_t alloc(T **p, size_t n, const char *name) {
        const size_t bytes = std::max<size_t>(n, 1) * sizeof(T), rz = redzone;
        uint8_t *raw = nullptr;
        hipError_t e = hipMalloc((void **)&raw, bytes + 2 * rz);
        if (e != hipSuccess) { *p = nullptr; return e; }
        if (rz) {
            e = hipMemset(raw, 0xA5, rz);
            if (e == hipSuccess) e = hipMemset(raw + rz + bytes, 0xA5, rz);
            if (e != hipSuccess) { hipFree(raw); *p = nullptr; return e; }
        }
        blocks.push_back({name, raw, bytes});
        *p = (T *)(raw + rz);
        return hipSuccess;
    }
    template <typename T>
    hipError_t release(T *&p) {
        if (!p) return hipSuccess;
        uint8_t *raw = (uint8_t *)p - redzone;
        for (size_t i = 0; i < blocks.size(); i++)
            if (blocks[i].raw == raw) { blocks.erase(blocks.begin() + i); break; }
        p = nullptr;
        return hipFree(raw);
    }
    void trace(const char *fmt, ...) {
        if (!trace_on) return;
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        fprintf(stderr, "[dne %d] %s\n", cfg.device_id, buf);
        fflush(stderr);
    }

    int fail(const char *fmt, ...) {
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        err = buf;
        return -1;
    }
    FwdArgs fwd(bool use_done) const {
        FwdArgs A;
        A.noise = noise; A.bases = bases; A.base_stride = base_stride;
        A.m_slot = m_slot; A.m_off = m_off; A.m_scale = m_scale; A.bn = bn; A.bn_mom = bn_mom;
        A.done = use_done ? done : nullptr; A.L = L;
        A.sub_sums = sub_now ? 1 : 0;
        if (tt_on) A.tt = tt; else A.tt.n = 0;
        return A;
    }
    EnvArgs env(int bc_mode) const {
        EnvArgs E;
        E.ram_prev = ram_prev; E.ram_cur = ram_cur; E.stacks = stacks; E.T = tables;
        E.ret = ret; E.sign = sign; E.step_reward = step_reward; E.len = len; E.done = done; E.stepped = stepped; E.action = action; E.step_counter = nullptr;
        E.bc = bc; E.bc_mode = bc ? bc_mode : 0; E.bc_max_steps = cfg.bc_max_steps; E.immortal = dbg_immortal;
        E.spec_prev = spec_prev; E.spec_cur = spec_cur; E.spec_rw = spec_rw; E.spec_stacks = spec_stacks; E.spec_y1 = spec_y1;
        E.tt_n = tt_on ? tt.n : 0;
        if (tt_on) memcpy(E.tt_member, tt.member, sizeof(E.tt_member));
        return E;
    }
    hipEvent_t event(size_t i) {
        while (ev_pool.size() <= i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            ev_pool.push_back(e);
        }
        return ev_pool[i];
    }
};

static int rec_reserve(dne_handle *h, int n_global, int per_world);

// the two small scratch arrays (scratch_f floats, scratch_i int64s) hold at least n elements each afterwards; contents are not kept
static int scratch_reserve(dne_handle *h, size_t n) {
    if (n <= h->scratch_cap) return 0;
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = h->release(h->scratch_f);
    if (e == hipSuccess) e = h->release(h->scratch_i);
    h->scratch_cap = 0;
    const size_t cap = std::max<size_t>(n, 65536);
    if (e == hipSuccess) e = h->alloc(&h->scratch_f, cap, "scratch_f");
    if (e == hipSuccess) e = h->alloc(&h->scratch_i, cap, "scratch_i");
    if (e != hipSuccess) return h->fail("scratch_reserve(%zu): %s", n, hipGetErrorString(e));
    h->scratch_cap = cap;
    return 0;
}

// Every entry point runs on the handle's device whatever the calling thread's current device is, and puts the
// caller's device back on the way out.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(const dne_handle *h) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != h->cfg.device_id) hipSetDevice(h->cfg.device_id); else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};

static void make_layout(int kind, int nact, Layout *L) {
    int o = 0;
    memset(L, 0, sizeof(*L));
    L->kind = kind; L->nact = nact;
    auto take = [&](int n) { int r = o; o += n; return r; };
    if (kind == DNE_KIND_ES) {   // creation order of trainable variables, policies.py:319-330
        L->c1w = take(4096); L->c1b = take(16); L->bn1b = take(16); L->bn1g = take(16);
        L->c2w = take(8192); L->c2b = take(32); L->bn2b = take(32); L->bn2g = take(32);
        L->fcw = take(3872 * 256); L->fcb = take(256); L->bn3b = take(256); L->bn3g = take(256);
        L->ow = take(256 * nact); L->ob = take(nact);
    } else if (kind == DNE_KIND_GA_LARGE) {   // models/dqn.py:39-47: variables in creation order (models/base.py:35-41)
        L->c1w = take(8 * 8 * 4 * 32); L->c1b = take(32); L->c2w = take(4 * 4 * 32 * 64); L->c2b = take(64);
        L->c3w = take(3 * 3 * 64 * 64); L->c3b = take(64);
        L->fcw = take(7744 * 512); L->fcb = take(512); L->ow = take(512 * nact); L->ob = take(nact);
        L->bn1b = L->bn1g = L->bn2b = L->bn2g = L->bn3b = L->bn3g = -1;
    } else {                     // policies.py:449-459 via tf_util.py:133-162
        L->c1w = take(4096); L->c1b = take(16); L->c2w = take(8192); L->c2b = take(32);
        L->fcw = take(3872 * 256); L->fcb = take(256); L->ow = take(256 * nact); L->ob = take(nact);
        L->bn1b = L->bn1g = L->bn2b = L->bn2g = L->bn3b = L->bn3g = -1;
    }
    L->P = o;
}

// Pillow Resample.c precompute_coeffs (BILINEAR, support 1.0) -- the filter WarpFrame uses
// (atari_wrappers.py:140-141); third-party algorithm restated from its published source.
static void pil_coeffs(int in_size, int out_size, int ksize, int *bounds, double *kk) {
    const double scale = (double)in_size / out_size;
    const double fscale = scale < 1.0 ? 1.0 : scale;
    const double support = fscale;
    for (int xx = 0; xx < out_size; xx++) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        if (xmax > ksize) { fprintf(stderr, "dne: a resize window of %d taps does not fit the table's %d\n", xmax, ksize); abort(); }
        double ww = 0.0;
        double *k = kk + xx * ksize;
        for (int x = 0; x < ksize; x++) k[x] = 0.0;
        for (int x = 0; x < xmax; x++) {
            double t = (x + xmin - center + 0.5) * (1.0 / fscale);
            t = t < 0 ? -t : t;
            k[x] = t < 1.0 ? 1.0 - t : 0.0;
            ww += k[x];
        }
        if (ww != 0.0)
            for (int x = 0; x < xmax; x++) k[x] /= ww;
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
}

static const uint8_t kPalette[16][3] = {
    {0, 0, 0},       {170, 170, 170}, {45, 50, 184},   {24, 26, 167},  {214, 214, 214}, {0, 28, 136},
    {236, 236, 236}, {84, 138, 210},  {198, 108, 58},  {181, 83, 40},  {192, 192, 192}, {252, 252, 84},
    {92, 186, 92},   {74, 74, 74},    {252, 144, 144}, {0, 44, 160}};

static void make_tables(ResizeLds *T) {
    // Fixed RS_KH / RS_KV taps per output pixel (the widest window PIL builds for these two scales): windows that would leave the frame are
    // shifted back inside and their weights shifted with them, zero weights filling the rest (0 + x*0.0 and acc + x*0.0 are exact no-ops).
    double kh[84 * RS_KH], kv[84 * RS_KV];
    int bh[84 * 2], bv[84 * 2];
    pil_coeffs(160, 84, RS_KH, bh, kh);
    pil_coeffs(210, 84, RS_KV, bv, kv);
    memset(T, 0, sizeof(*T));
    for (int xx = 0; xx < 84; xx++) {
        const int x0 = bh[2 * xx], d = x0 + RS_KH > 160 ? x0 + RS_KH - 160 : 0;
        T->xmin[xx] = (uint8_t)(x0 - d);
        for (int t = 0; t < RS_KH; t++) T->kh[xx * RS_KH + t] = t >= d ? kh[xx * RS_KH + t - d] : 0.0;
    }
    for (int yy = 0; yy < 84; yy++) {
        const int y0 = bv[2 * yy], d = y0 + RS_KV > 210 ? y0 + RS_KV - 210 : 0;
        T->ymin[yy] = (uint8_t)(y0 - d);
        for (int t = 0; t < RS_KV; t++) T->kv[yy * RS_KV + t] = t >= d ? kv[yy * RS_KV + t - d] : 0.0;
    }
    for (int a = 0; a < 16; a++)
        for (int b = 0; b < 16; b++) {   // MaxAndSkip max (atari_wrappers.py:105) then WarpFrame gray (:139)
            const uint8_t r = std::max(kPalette[a][0], kPalette[b][0]);
            const uint8_t g = std::max(kPalette[a][1], kPalette[b][1]);
            const uint8_t bl = std::max(kPalette[a][2], kPalette[b][2]);
            volatile float t0 = (float)r * 0.299f, t1 = (float)g * 0.587f, t2 = (float)bl * 0.114f;
            volatile float s = t0 + t1;
            T->gray2[(a << 4) | b] = s + t2;
        }
}

template <typename T>
static hipError_t dalloc(T **p, size_t n) {
    return hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T));
}

// scratch device buffer of one call: freed on every exit path
template <typename T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t n) { return dalloc(&p, n); }
    operator T *() const { return p; }
};

extern "C" int dne_num_params(int kind, int nact) {
    Layout L;
    make_layout(kind, nact, &L);
    return L.P;
}

extern "C" const char *dne_last_error(dne_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

constexpr size_t OVERFETCH_FLOATS = 2 * 8 * 256 + 64;   // one row block of the streaming fc + the 64 floats of slack the table always had
static int grow_bases(dne_handle *h, int cap) {
    if (cap <= h->base_cap) return 0;
    float *nb = nullptr;
    HCHECK(h, h->alloc(&nb, (size_t)cap * h->base_stride + OVERFETCH_FLOATS, "bases"));   // (the same over-fetch behind the last slot)
    HCHECK(h, hipMemsetAsync(nb + (size_t)cap * h->base_stride, 0, OVERFETCH_FLOATS * sizeof(float), h->stream));
    if (h->bases) {
        HCHECK(h, hipMemcpyAsync(nb, h->bases, (size_t)h->base_cap * h->base_stride * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        HCHECK(h, hipStreamSynchronize(h->stream));
        HCHECK(h, h->release(h->bases));
    }
    for (int s = cap - 1; s >= std::max(h->base_cap, 1); s--) h->free_slots.push_back(s);
    h->bases = nb;
    h->base_cap = cap;
    return 0;
}

extern "C" int dne_create(const dne_config *cfg, dne_handle **out) {
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        g_create_error = "dne_create: no HIP device visible (this engine has no CPU fallback)";
        return -1;
    }
    if (cfg->max_members <= 0 || cfg->n_actions <= 1 || cfg->n_actions > 32 ||
        (cfg->policy_kind != DNE_KIND_ES && cfg->policy_kind != DNE_KIND_GA && cfg->policy_kind != DNE_KIND_GA_LARGE)) {
        g_create_error = "dne_create: bad config";
        return -1;
    }
    dne_handle *h = new dne_handle();
    h->cfg = *cfg;
    auto bail = [&](int) { g_create_error = h->err; delete h; return -1; };
#define CCHECK(expr) do { if ((expr) != 0) return bail(0); } while (0)
#define CH(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { h->fail("%s -> %s", #expr, hipGetErrorString(_e)); return bail(0); } } while (0)
    CH(hipSetDevice(cfg->device_id));
    CH(hipStreamCreate(&h->stream));
    h->sub_streams.push_back(h->stream);
    // tuning knobs (measurement only; every value is clamped to what the kernels support)
    auto env_int = [](const char *name, int lo, int hi, int *dst) {
        if (const char *e = getenv(name)) *dst = std::max(lo, std::min(hi, atoi(e)));
    };
    auto wg_size = [](int v) { return v >= 1024 ? 1024 : v >= 512 ? 512 : 256; };   // the renderer needs >= 210 threads, whole waves
    {
        int rz = 1, tr = 0, ds = 0;
        env_int("DNE_REDZONE", 0, 1, &rz); env_int("DNE_TRACE", 0, 1, &tr); env_int("DNE_DEBUG_SYNC", 0, 1, &ds);
        h->redzone = rz ? 4096 : 0; h->trace_on = tr != 0; h->debug_sync = ds != 0;
        int sc = 1;
        env_int("DNE_STAGED_COPY", 0, 1, &sc);
        h->staged_copies = sc != 0;
    }
    env_int("DNE_GA_SORT", 0, 1, &h->ga_sort);
    env_int("DNE_CONV_FUSED", 0, 1, &h->conv_fused);
    env_int("DNE_CONV_FUSED_MIN", 1, 1 << 20, &h->conv_fused_min);
    CH(hipFuncSetAttribute((const void *)k_conv12<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Conv12Lds)));
    CH(hipFuncSetAttribute((const void *)k_conv12<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Conv12Lds)));
    CH(hipFuncSetAttribute((const void *)k_conv12t<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Conv12Lds)));
    CH(hipFuncSetAttribute((const void *)k_conv12t<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Conv12Lds)));
    CH(hipFuncSetAttribute((const void *)k_unit_order, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CH(hipFuncSetAttribute((const void *)k_fc_ref<4>, hipFuncAttributeMaxDynamicSharedMemorySize, FCREF4_LDS));   // the running fold
    CH(hipFuncSetAttribute((const void *)k_lconv_mfma<32, 64, 4, 2, 21, 11, 1, 34, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lconv_mfma_lds_bytes<32, 4, 2, 11, 34>()));
    CH(hipFuncSetAttribute((const void *)k_lconv_mfma<64, 64, 3, 1, 11, 11, 1, 68, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lconv_mfma_lds_bytes<64, 3, 1, 11, 68>()));
    CH(hipFuncSetAttribute((const void *)k_lconv_mfma<32, 64, 4, 2, 21, 11, 1, 34, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lconv_mfma_lds_bytes<32, 4, 2, 11, 34>()));
    CH(hipFuncSetAttribute((const void *)k_lconv_mfma<64, 64, 3, 1, 11, 11, 1, 68, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lconv_mfma_lds_bytes<64, 3, 1, 11, 68>()));
    CH(hipFuncSetAttribute((const void *)k_out<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));   // the DNE_OUT_LDS_KB reservation
    CH(hipFuncSetAttribute((const void *)k_out<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CH(hipFuncSetAttribute((const void *)k_out<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CH(hipFuncSetAttribute((const void *)k_out<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    env_int("DNE_NSUB", 1, 4, &h->nsub_fixed);
    env_int("DNE_NSUB_FULL", 1, 4, &h->nsub_full);
    env_int("DNE_NSUB_MID", 1, 4, &h->nsub_mid);
    env_int("DNE_FC_TAIL_MAX", 1, 1 << 20, &h->fc_tail_max);
    env_int("DNE_FC_QUAD_MAX", 0, 1 << 20, &h->fc_quad_max);
    env_int("DNE_FC_TAILK_MAX", 0, 1 << 20, &h->fc_tailk_max);
    env_int("DNE_DEBUG_SKIP", 0, 7, &h->dbg_skip);
    env_int("DNE_DEBUG_IMMORTAL", 0, 1, &h->dbg_immortal);
#ifdef DNE_PHASE_CLOCK
    {
        int burn = 0;
        env_int("DNE_ENV_BURN", 0, 1 << 20, &burn);
        CH(hipMemcpyToSymbol(HIP_SYMBOL(g_env_burn), &burn, sizeof(int)));
    }
#endif
    env_int("DNE_TAIL_TABLE", 0, 1, &h->tt_enable);
    env_int("DNE_HEAD_THREADS", 256, 320, &h->head_threads); h->head_threads = h->head_threads >= 320 ? 320 : 256;
    env_int("DNE_RENDER_THREADS", 256, 1024, &h->render_threads); h->render_threads = wg_size(h->render_threads);
    env_int("DNE_BAND_THREADS", 256, 1024, &h->band_threads); h->band_threads = wg_size(h->band_threads);
    env_int("DNE_TAIL_FUSED_MAX", 0, 1 << 20, &h->tail_fused_max);
    env_int("DNE_FC_PAIRS", 1, 2, &h->fc_pairs);
    env_int("DNE_CONV1_FPW", 1, 8, &h->conv1_fpw);
    env_int("DNE_CONV_SPLIT_MAX", 0, 1 << 20, &h->conv_split_max);
    env_int("DNE_CONV12T_MAX", 0, 1 << 20, &h->conv12t_max);
    env_int("DNE_CONV_SPLIT_MID", 0, 1 << 20, &h->conv_split_mid);
    env_int("DNE_FC2_MIN", 2, 1 << 30, &h->fc2_min_total);
    env_int("DNE_SPEC_MAX", 0, 64, &h->spec_max);
    env_int("DNE_SPEC_BANDS", 1, 12, &h->spec_bands);
    env_int("DNE_SPEC_CONV1", 0, 1, &h->spec_conv1);
    env_int("DNE_FC_DUO", 0, 1, &h->fc_duo);
    env_int("DNE_DUO_LAG", 0, 64, &h->duo_lag);
    env_int("DNE_FC_PRIO", 0, 3, &h->fc_prio);
    env_int("DNE_DUO_SWEEP", 0, 2, &h->duo_sweep);
    env_int("DNE_DUO_SYNC", 1, 8, &h->duo_sync);
    env_int("DNE_DUO_FAT", 0, 1, &h->duo_fat); env_int("DNE_FC_RING", 0, 2, &h->ring_on); env_int("DNE_RING_MIN", 0, 1 << 30, &h->ring_min); env_int("DNE_RING_PRE", 0, 1, &h->ring_pre); env_int("DNE_SUB_RENDER_FUSED", 0, 1, &h->sub_render_fused);
    env_int("DNE_BURST", 1, 256, &h->burst);
    env_int("DNE_BURST_TAIL", 1, 256, &h->burst_tail);
    env_int("DNE_DUO_GRID", 0, 1 << 16, &h->duo_grid);
    // measured (tools/ga_lockstep_profile.py, tools/ab_inproc.py --pairs 312 / 625): Deep-GA children 97 .. 320
    // alive, two windows, a grid of 512 workgroups at wave priority 3; ES pairs 97 .. 450 alive, three windows, the whole launch
    // resident, no priority (its chain of small kernels must not starve); above those widths the streaming kernels win
    if (cfg->policy_kind == DNE_KIND_ES) { h->fc_sub = 2; h->fc_sub_max = 450; h->fc_sub_nsub = 3; h->fc_sub_grid = 1 << 20; h->fc_sub_prio = 0; }
    else { h->fc_sub = 1; h->fc_sub_max = 320; h->fc_sub_nsub = 4; h->fc_sub_grid = 512; h->fc_sub_prio = 0; }   // round 6: four windows at wave priority 0 (was two at 3): Deep GA 1.02-1.05 vs 0.98-1.00 M env-steps/s same-box; five / six windows 0.91-0.94
    env_int("DNE_FC_SUB", 0, 2, &h->fc_sub);
    env_int("DNE_FC_SUB_MIN", 1, 1 << 30, &h->fc_sub_min);
    env_int("DNE_FC_SUB_MAX", 1, 1 << 30, &h->fc_sub_max);
    env_int("DNE_FC_SUB_NSUB", 1, 4, &h->fc_sub_nsub);
    env_int("DNE_FC_SUB_SPW", 0, 8, &h->fc_sub_spw);
    env_int("DNE_FC_SUB_PRIO", 0, 3, &h->fc_sub_prio);
    env_int("DNE_FC_SUB_GRID", 1, 1 << 20, &h->fc_sub_grid);
    env_int("DNE_FC_SUB_HEAD", 0, 1, &h->fc_sub_head);
    if (h->fc_sub_spw != 1 && h->fc_sub_spw != 2 && h->fc_sub_spw != 4 && h->fc_sub_spw != 8) h->fc_sub_spw = 0;
    env_int("DNE_CONV2_REF_FPW", 1, 8, &h->conv2_ref_fpw);
    env_int("DNE_DUO_SOLO_BELOW", 0, 1 << 30, &h->duo_solo_below);
    env_int("DNE_DUO_HEAD_FUSED", 0, 1, &h->duo_head_fused);
    env_int("DNE_OUT_LDS_KB", 0, 64, &h->out_lds_kb);
    env_int("DNE_FC_DUO_MIN", 2, 1 << 30, &h->fc_duo_min);
    env_int("DNE_RENDER_BANDS", 1, 84, &h->render_bands);
    env_int("DNE_RENDER_WG_MAX", 1, 1 << 20, &h->render_wg_max);
    env_int("DNE_FC_CHAIN_MIN", 1, 1 << 30, &h->fc_chain_min);
    env_int("DNE_FC_RB", 2, 8, &h->fc_rb);
    env_int("DNE_FC_GRID", 1, 1 << 16, &h->fc_grid);
    for (int s = 1; s < 4; s++) { hipStream_t st; CH(hipStreamCreate(&st)); h->sub_streams.push_back(st); }   // more than four windows measured slower (round 3, 5 / 6 / 8 at full width: +7.5 / +5.4 / +7.2 %; round 4 with one k_fc_duo workgroup per CU: +8.4 / +5.9 / +4.4 %)
    make_layout(cfg->policy_kind, cfg->n_actions, &h->L);
    h->M = cfg->max_members;
    h->F = cfg->policy_kind == DNE_KIND_ES ? (cfg->ref_count > 0 ? cfg->ref_count : 128) : 0;
    if (h->F % 8) { h->fail("ref_count must be a multiple of 8"); return bail(0); }
    h->ref_chunk = cfg->ref_chunk > 0 ? cfg->ref_chunk : 512;
    h->ref_chunk = std::min(h->ref_chunk, h->M);
    h->base_stride = ((size_t)h->L.P + 63) / 64 * 64;
    const size_t M = h->M;
    CCHECK(grow_bases(h, 1));
    CH(hipMemset(h->bases, 0, h->base_stride * sizeof(float)));
    CH(h->alloc(&h->opt_m, h->L.P, "opt_m")); CH(h->alloc(&h->opt_v, h->L.P, "opt_v")); CH(h->alloc(&h->g, h->L.P, "g"));
    CH(hipMemset(h->opt_m, 0, h->L.P * sizeof(float))); CH(hipMemset(h->opt_v, 0, h->L.P * sizeof(float)));
    CH(h->alloc(&h->partial, 2 * ((size_t)h->L.P / 256 + 1), "partial"));
    if (h->F) CH(h->alloc(&h->ref, (size_t)h->F * OB_BYTES, "ref"));
    if (h->F) CH(h->alloc(&h->ref_f32, (size_t)h->F * RF_FRAME, "ref_f32"));
    env_int("DNE_CONV1_SHARED", 0, 1, &h->conv1_shared);
    CH(hipFuncSetAttribute((const void *)k_conv1_ref_shared<16>, hipFuncAttributeMaxDynamicSharedMemorySize, RF_FRAME * (int)sizeof(float)));
    CH(hipFuncSetAttribute((const void *)k_conv1_ref_shared<8>, hipFuncAttributeMaxDynamicSharedMemorySize, RF_FRAME * (int)sizeof(float)));
    CH(h->alloc(&h->m_slot, M, "m_slot")); CH(h->alloc(&h->m_off, M, "m_off")); CH(h->alloc(&h->m_scale, M, "m_scale"));
    CH(hipMemset(h->m_slot, 0, M * sizeof(int32_t))); CH(hipMemset(h->m_off, 0, M * sizeof(int64_t)));
    CH(hipMemset(h->m_scale, 0, M * sizeof(float)));
    CH(h->alloc(&h->bn, M * 608, "bn")); CH(h->alloc(&h->bn_mom, M * 608, "bn_mom"));
    CH(h->alloc(&h->ram_prev, M * 128, "ram_prev")); CH(h->alloc(&h->ram_cur, M * 128, "ram_cur")); CH(h->alloc(&h->stacks, M * OB_BYTES, "stacks"));
    CH(hipMemset(h->stacks, 0, M * OB_BYTES));
    CH(h->alloc(&h->tables, 1, "tables"));
    {
        ResizeLds T;
        make_tables(&T);
        CH(hipMemcpy(h->tables, &T, sizeof(T), hipMemcpyHostToDevice));
    }
    CH(h->alloc(&h->ret, M, "ret")); CH(h->alloc(&h->sign, M, "sign")); CH(h->alloc(&h->step_reward, M, "step_reward"));
    CH(h->alloc(&h->logits, M * cfg->n_actions, "logits"));
    CH(h->alloc(&h->len, M, "len")); CH(h->alloc(&h->done, M, "done")); CH(h->alloc(&h->action, M, "action")); CH(h->alloc(&h->seeds, M, "seeds")); CH(h->alloc(&h->stepped, M, "stepped"));
    CH(hipMemset(h->done, 0, M * sizeof(int32_t))); CH(hipMemset(h->len, 0, M * sizeof(int32_t)));
    h->large = cfg->policy_kind == DNE_KIND_GA_LARGE;
    h->ga_materialize = cfg->policy_kind == DNE_KIND_ES ? 0 : 1;
    env_int("DNE_GA_MATERIALIZE", 0, 1, &h->ga_materialize);
    env_int("DNE_LFC_COLS_MAX", 0, 1 << 20, &h->lfc_cols_max);
    env_int("DNE_LFC_PAD", 0, 2, &h->lfc_pad);
    if (h->large) h->fc_rb = 8;      // the streamed LargeModel fc: 8-row batches measured 8 % faster than 4
    env_int("DNE_FC_RB", 2, 8, &h->fc_rb);
    if (cfg->policy_kind == DNE_KIND_ES) h->ga_materialize = 0;   // ES members are antithetic pairs over one theta: nothing to write out
    if (h->large) { CH(h->alloc(&h->y1, M * 14112, "y1")); CH(h->alloc(&h->y2, M * 7744, "y2")); CH(h->alloc(&h->y3, M * 7744, "y3")); CH(h->alloc(&h->y3t, M * 512, "y3t")); }
    else { CH(h->alloc(&h->y1, M * 7056, "y1")); CH(h->alloc(&h->y2, M * 3872 + 64, "y2"));   /* (+ 64: k_fc_ring fetches the eight activations behind a slice's end and never uses them) */ CH(h->alloc(&h->y3, M * 256, "y3")); CH(h->alloc(&h->y3t, M * 4 * 256, "y3t")); }
    CH(h->alloc(&h->unit_order, M * 4, "unit_order"));
    if (!h->large && h->fc_sub) CH(h->alloc(&h->y3s, M * 32 * 256, "y3s"));
    if (!h->large && h->ring_on && cfg->policy_kind == DNE_KIND_ES) CH(h->alloc(&h->theta_perm, (size_t)(3872 + 16) * 256, "theta_perm"));
    if (cfg->n_actions > SPEC_ACTIONS - 2) h->spec_max = 0;
    if (h->spec_max > 0) {   // candidate outcomes of the speculative tail: [list position][action]
        const size_t rows = (size_t)h->spec_max * SPEC_ACTIONS;
        CH(h->alloc(&h->spec_prev, rows * 128, "spec_prev")); CH(h->alloc(&h->spec_cur, rows * 128, "spec_cur"));
        CH(h->alloc(&h->spec_rw, rows * 2, "spec_rw")); CH(h->alloc(&h->spec_stacks, rows * OB_BYTES, "spec_stacks"));
        CH(h->alloc(&h->spec_y1, rows * 7056, "spec_y1"));
        CH(hipMemset(h->spec_prev, 0, rows * 128)); CH(hipMemset(h->spec_cur, 0, rows * 128));   // RAM bytes past the live 40 stay zero
    }
    if (h->F) {
        const size_t rr = (size_t)h->ref_chunk * h->F;
        for (int w = 0; w < 2; w++) {
            CH(h->alloc(&h->y1r[w], rr * 7056, "y1r[w]")); CH(h->alloc(&h->y2r[w], rr * Y2_PAD_ROW, "y2r[w]")); CH(h->alloc(&h->y3pr[w], rr * 4 * 256, "y3pr[w]"));
            CH(h->alloc(&h->fr1[w], rr * 2 * 16, "fr1[w]")); CH(h->alloc(&h->fr2[w], rr * 2 * 32, "fr2[w]"));
            CH(hipEventCreateWithFlags(&h->ev_ref[w], hipEventDisableTiming));
        }
    }
    CH(h->alloc(&h->list_a, M, "list_a")); CH(h->alloc(&h->list_b, M, "list_b")); CH(h->alloc(&h->count_dev, 8 + TT_MAX, "count_dev"));
    CH(hipHostMalloc((void **)&h->count_host, (8 + TT_MAX) * sizeof(int), hipHostMallocDefault));
    if (cfg->record_bc) {
        h->bc_bytes = cfg->policy_kind == DNE_KIND_ES && !cfg->bc_final_only ? M * (size_t)std::max(cfg->bc_max_steps, 1) * 128 : M * 128;
        CH(h->alloc(&h->bc, h->bc_bytes, "bc"));
        CH(hipMemset(h->bc, 0, h->bc_bytes));   // the emulator writes the RAM_LIVE bytes of a row; the other bytes of the 128 stay zero for good
    }
    h->scratch_cap = std::max<size_t>(4 * M + 64, 65536);
    CH(h->alloc(&h->scratch_f, h->scratch_cap, "scratch_f")); CH(h->alloc(&h->scratch_i, h->scratch_cap, "scratch_i"));
    CH(hipEventCreate(&h->ev_a)); CH(hipEventCreate(&h->ev_b));
    for (int i = 0; i < 64; i++) { hipEvent_t ev; CH(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); h->fc_ring.push_back(ev); }
    CH(hipDeviceSynchronize());
#undef CH
#undef CCHECK
    {
        size_t tot = 0;
        for (auto &b : h->blocks) tot += b.bytes;
        h->trace("engine created: kind %d, %d members, %d reference frames, %zu device buffers, %.2f GB", cfg->policy_kind, h->M,
                 h->F, h->blocks.size(), tot / 1e9);
    }
    *out = h;
    return 0;
}

// Every device buffer sits between two 4 KiB zones filled with 0xA5 at allocation; a kernel that writes outside its
// buffer shows up here (reads cannot be seen).  Returns the number of damaged zones (0 = clean), -1 on a HIP error;
// dne_last_error names the first damaged buffer.
extern "C" int dne_check_redzones(dne_handle *h) {
    DeviceGuard dg(h);
    if (!h->redzone) return 0;
    HCHECK(h, hipDeviceSynchronize());
    const size_t rz = h->redzone;
    std::vector<uint8_t> host(2 * rz);
    int bad = 0;
    for (auto &b : h->blocks) {
        HCHECK(h, hipMemcpy(host.data(), b.raw, rz, hipMemcpyDeviceToHost));
        HCHECK(h, hipMemcpy(host.data() + rz, b.raw + rz + b.bytes, rz, hipMemcpyDeviceToHost));
        for (int side = 0; side < 2; side++) {
            size_t first = rz;
            for (size_t i = 0; i < rz; i++)
                if (host[side * rz + i] != 0xA5) { first = i; break; }
            if (first < rz) {
                if (!bad) h->fail("red zone %s buffer '%s' (%zu bytes) damaged at byte %zu", side ? "after" : "before", b.name.c_str(),
                                  b.bytes, side ? first : rz - first);
                bad++;
            }
        }
    }
    return bad;
}

static void comm_destroy(dne_handle *h);

extern "C" void dne_destroy(dne_handle *h) {
    if (!h) return;
    DeviceGuard dg(h);
    hipDeviceSynchronize();
    if (dne_check_redzones(h) > 0) fprintf(stderr, "libdne_hip: %s\n", h->err.c_str());
    comm_destroy(h);
    for (auto &b : h->blocks) hipFree(b.raw);
    h->blocks.clear();
    for (int i = 0; i < 2; i++) {
        if (h->stage_buf[i]) hipHostFree(h->stage_buf[i]);
        if (h->stage_ev[i]) hipEventDestroy(h->stage_ev[i]);
    }
    for (hipEvent_t e : h->ev_pool) hipEventDestroy(e);
    if (h->count_host) hipHostFree(h->count_host);
    for (hipEvent_t e : h->fc_ring) hipEventDestroy(e);
    for (hipEvent_t e : h->ev_ref) if (e) hipEventDestroy(e);
    if (h->ev_a) hipEventDestroy(h->ev_a);
    if (h->ev_b) hipEventDestroy(h->ev_b);
    for (size_t s = 1; s < h->sub_streams.size(); s++) hipStreamDestroy(h->sub_streams[s]);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int dne_get_profile(dne_handle *h, dne_profile *out) {
    *out = h->prof;
    return 0;
}

// ------------------------------------------------------------------------------- noise / theta
// Large transfers go through two engine-owned pinned buffers instead of handing the caller's pageable memory to the
// runtime (which, above 128 MiB, registers the caller's pages with the GPU and copies from them in place -- the only
// transfer of that kind in a generation's life is the 1 GB table upload).  Small copies keep using hipMemcpy.
constexpr size_t STAGE_BYTES = 32u << 20, STAGE_MIN = 64u << 20;

static int stage_init(dne_handle *h) {
    for (int i = 0; i < 2; i++) {
        if (!h->stage_buf[i]) HCHECK(h, hipHostMalloc((void **)&h->stage_buf[i], STAGE_BYTES, hipHostMallocDefault));
        if (!h->stage_ev[i]) HCHECK(h, hipEventCreateWithFlags(&h->stage_ev[i], hipEventDisableTiming));
    }
    return 0;
}

static int copy_h2d(dne_handle *h, void *dst, const void *src, size_t bytes) {
    if (bytes < STAGE_MIN || !h->staged_copies) { HCHECK(h, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return 0; }
    if (stage_init(h)) return -1;
    size_t done = 0;
    for (int c = 0; done < bytes; c++) {
        const int b = c & 1;
        const size_t n = std::min(STAGE_BYTES, bytes - done);
        if (c >= 2) HCHECK(h, hipEventSynchronize(h->stage_ev[b]));   // the copy that last read this buffer
        memcpy(h->stage_buf[b], (const uint8_t *)src + done, n);
        HCHECK(h, hipMemcpyAsync((uint8_t *)dst + done, h->stage_buf[b], n, hipMemcpyHostToDevice, h->stream));
        HCHECK(h, hipEventRecord(h->stage_ev[b], h->stream));
        done += n;
    }
    HCHECK(h, hipStreamSynchronize(h->stream));
    return 0;
}

static int copy_d2h(dne_handle *h, void *dst, const void *src, size_t bytes) {
    if (bytes < STAGE_MIN || !h->staged_copies) { HCHECK(h, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return 0; }
    if (stage_init(h)) return -1;
    HCHECK(h, hipStreamSynchronize(h->stream));
    size_t issued = 0, drained = 0;
    for (int c = 0; drained < bytes; c++) {   // chunk c is copied out of its buffer while chunk c + 1 arrives in the other
        const int b = c & 1;
        if (c == 0) {
            const size_t n = std::min(STAGE_BYTES, bytes);
            HCHECK(h, hipMemcpyAsync(h->stage_buf[0], src, n, hipMemcpyDeviceToHost, h->stream));
            HCHECK(h, hipEventRecord(h->stage_ev[0], h->stream));
            issued = n;
        }
        const size_t mine = std::min(STAGE_BYTES, bytes - drained);
        if (issued < bytes) {
            const size_t n = std::min(STAGE_BYTES, bytes - issued);
            HCHECK(h, hipMemcpyAsync(h->stage_buf[b ^ 1], (const uint8_t *)src + issued, n, hipMemcpyDeviceToHost, h->stream));
            HCHECK(h, hipEventRecord(h->stage_ev[b ^ 1], h->stream));
            issued += n;
        }
        HCHECK(h, hipEventSynchronize(h->stage_ev[b]));
        memcpy((uint8_t *)dst + drained, h->stage_buf[b], mine);
        drained += mine;
    }
    return 0;
}

extern "C" int dne_noise_alloc(dne_handle *h, size_t count) {
    DeviceGuard dg(h);
    if (count == 0) return h->fail("dne_noise_alloc: empty table");
    HCHECK(h, h->release(h->noise));
    HCHECK(h, h->release(h->noise_pre));
    h->noise_count = 0; h->noise_pre_count = 0;
    // the streaming fc kernels (k_fc_duo, k_fc_sub) fetch one 8-row block past a unit's last row and never use it: for small action
    // counts that block ends behind the member's parameter slice, i.e. up to OVERFETCH_FLOATS behind the table's last legal slice
    HCHECK(h, h->alloc(&h->noise, count + OVERFETCH_FLOATS, "noise"));
    HCHECK(h, hipMemset(h->noise + count, 0, OVERFETCH_FLOATS * sizeof(float)));
    h->noise_count = count;
    h->trace("noise table allocated: %zu floats", count);
    return 0;
}

extern "C" int dne_noise_write(dne_handle *h, size_t offset, const float *host, size_t count) {
    DeviceGuard dg(h);
    if (!h->noise || offset + count > h->noise_count) return h->fail("dne_noise_write: [%zu, %zu) outside the table of %zu", offset, offset + count, h->noise_count);
    h->noise_pre_count = 0;   // the scaled copy is stale
    return copy_h2d(h, h->noise + offset, host, count * sizeof(float));
}

extern "C" int dne_noise_upload(dne_handle *h, const float *host, size_t count) {
    if (dne_noise_alloc(h, count)) return -1;
    if (dne_noise_write(h, 0, host, count)) return -1;
    h->trace("noise table uploaded");
    return 0;
}

static int check_noise_range(dne_handle *h, int64_t idx, int64_t dim) {
    if (!h->noise) return h->fail("noise table not uploaded (dne_noise_upload)");
    if (idx < 0 || (uint64_t)(idx + dim) > h->noise_count) return h->fail("noise index %lld + %lld outside the table of %zu", (long long)idx, (long long)dim, h->noise_count);
    return 0;
}

extern "C" int dne_noise_get(dne_handle *h, int64_t idx, int dim, float *out) {
    DeviceGuard dg(h);
    if (check_noise_range(h, idx, dim)) return -1;
    HCHECK(h, hipMemcpy(out, h->noise + idx, (size_t)dim * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int dne_set_theta(dne_handle *h, int slot, const float *theta, size_t n) {
    DeviceGuard dg(h);
    if (n != (size_t)h->L.P) return h->fail("dne_set_theta: expected %d parameters, got %zu", h->L.P, n);
    if (slot < 0) return h->fail("bad slot");
    if (grow_bases(h, slot + 1)) return -1;
    HCHECK(h, hipMemcpy(h->bases + (size_t)slot * h->base_stride, theta, n * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int dne_get_theta(dne_handle *h, int slot, float *out, size_t n) {
    DeviceGuard dg(h);
    if (n != (size_t)h->L.P) return h->fail("dne_get_theta: expected %d parameters, got %zu", h->L.P, n);
    if (slot < 0 || slot >= h->base_cap) return h->fail("bad slot");
    HCHECK(h, hipStreamSynchronize(h->stream));
    HCHECK(h, hipMemcpy(out, h->bases + (size_t)slot * h->base_stride, n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int dne_set_ref_batch(dne_handle *h, const uint8_t *ref, int count) {
    DeviceGuard dg(h);
    if (h->L.kind != DNE_KIND_ES) return h->fail("reference batch is an ESAtariPolicy concept");
    if (count != h->F) return h->fail("dne_set_ref_batch: engine was created for %d reference frames, got %d", h->F, count);
    HCHECK(h, hipMemcpy(h->ref, ref, (size_t)count * OB_BYTES, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_ref_to_float, dim3((count * RF_FRAME + 255) / 256), dim3(256), 0, h->stream, (const uint8_t *)h->ref, count, h->ref_f32);
    HCHECK(h, hipGetLastError());
    HCHECK(h, hipStreamSynchronize(h->stream));
    h->ref_set = true;
    return 0;
}

extern "C" int dne_materialize(dne_handle *h, const int64_t *idx, int n, float sigma, float *out_host) {
    DeviceGuard dg(h);
    for (int i = 0; i < n; i++)
        if (check_noise_range(h, idx[i], h->L.P)) return -1;
    const size_t need = (size_t)n * 2 * h->L.P;
    if (need > h->mat_cap) {
        HCHECK(h, h->release(h->mat_out));
        HCHECK(h, h->alloc(&h->mat_out, need, "mat_out"));
        h->mat_cap = need;
    }
    if (scratch_reserve(h, (size_t)n)) return -1;
    HCHECK(h, hipMemcpyAsync(h->scratch_i, idx, n * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HCHECK(h, hipEventRecord(h->ev_a, h->stream));
    hipLaunchKernelGGL(k_materialize, dim3((h->L.P + 255) / 256, n), dim3(256), 0, h->stream, h->bases, h->noise,
                       h->scratch_i, h->L.P, sigma, h->mat_out);
    HCHECK(h, hipEventRecord(h->ev_b, h->stream));
    HCHECK(h, hipStreamSynchronize(h->stream));
    float ms = 0;
    HCHECK(h, hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
    h->prof.materialize_ms = ms;
    if (out_host && copy_d2h(h, out_host, h->mat_out, need * sizeof(float))) return -1;
    return 0;
}

static void launch_env_reset(dne_handle *h, int n) {
    const EnvArgs E = h->env(0);
    hipLaunchKernelGGL(k_env_reset_logic, dim3((n + 63) / 64), dim3(64), 0, h->stream, E, (const uint32_t *)h->seeds, n);
    hipLaunchKernelGGL(k_env_render, dim3(n), dim3(256), 0, h->stream, E, (const int *)nullptr, 1, 1, 1);   // FrameStack reset: 4 copies
}

static void launch_env_step(dne_handle *h, const EnvArgs &E, const int *list, int count, int gsize, int tslimit,
                            hipStream_t st = nullptr) {
    if (!st) st = h->stream;
    const int items = count * gsize;
    hipLaunchKernelGGL(k_env_logic, dim3((items + 63) / 64), dim3(64), 0, st, E, list, gsize, items, tslimit);
    if (h->dbg_skip & 4) return;
    hipLaunchKernelGGL(k_env_render, dim3(items), dim3(items <= 192 ? 1024 : h->render_threads), 0, st, E, list, gsize, 0, 1);
}

// ------------------------------------------------------------------------------- env ABI
static int check_n(dne_handle *h, int n) {
    if (n <= 0 || n > h->M) return h->fail("n = %d outside [1, max_members = %d]", n, h->M);
    return 0;
}

extern "C" int dne_env_reset(dne_handle *h, int n, const uint32_t *seeds) {
    DeviceGuard dg(h);
    if (check_n(h, n)) return -1;
    HCHECK(h, hipMemcpyAsync(h->seeds, seeds, n * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    launch_env_reset(h, n);
    HCHECK(h, hipGetLastError());
    HCHECK(h, hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int dne_env_step(dne_handle *h, int n, const int32_t *actions, float *reward, int32_t *done) {
    DeviceGuard dg(h);
    if (check_n(h, n)) return -1;
    for (int i = 0; i < n; i++)
        if (actions[i] < 0 || actions[i] >= h->cfg.n_actions) return h->fail("action %d out of range", actions[i]);
    HCHECK(h, hipMemcpyAsync(h->action, actions, n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HCHECK(h, hipMemsetAsync(h->step_reward, 0, n * sizeof(float), h->stream));
    launch_env_step(h, h->env(0), nullptr, n, 1, 0x7fffffff);
    HCHECK(h, hipGetLastError());
    if (reward) HCHECK(h, hipMemcpyAsync(reward, h->step_reward, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (done) HCHECK(h, hipMemcpyAsync(done, h->done, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HCHECK(h, hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int dne_env_observation(dne_handle *h, int n, uint8_t *out) {
    DeviceGuard dg(h);
    if (check_n(h, n)) return -1;
    HCHECK(h, hipStreamSynchronize(h->stream));
    HCHECK(h, hipMemcpy(out, h->stacks, (size_t)n * OB_BYTES, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int dne_env_ram(dne_handle *h, int n, uint8_t *out) {
    DeviceGuard dg(h);
    if (check_n(h, n)) return -1;
    HCHECK(h, hipStreamSynchronize(h->stream));
    HCHECK(h, hipMemcpy(out, h->ram_cur, (size_t)n * 128, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int dne_env_set_observation(dne_handle *h, int n, const uint8_t *obs) {
    DeviceGuard dg(h);
    if (check_n(h, n)) return -1;
    HCHECK(h, hipMemcpy(h->stacks, obs, (size_t)n * OB_BYTES, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int dne_env_set_ram(dne_handle *h, int n, const uint8_t *ram_prev, const uint8_t *ram_cur) {
    DeviceGuard dg(h);
    if (check_n(h, n)) return -1;
    HCHECK(h, hipMemcpyAsync(h->ram_prev, ram_prev, (size_t)n * 128, hipMemcpyHostToDevice, h->stream));
    HCHECK(h, hipMemcpyAsync(h->ram_cur, ram_cur, (size_t)n * 128, hipMemcpyHostToDevice, h->stream));
    std::vector<int32_t> ones(n, 1);
    HCHECK(h, hipMemcpyAsync(h->stepped, ones.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    const EnvArgs E = h->env(0);
    hipLaunchKernelGGL(k_env_render, dim3(n), dim3(256), 0, h->stream, E, (const int *)nullptr, 1, 1, 1);   // fill: 4 copies
    HCHECK(h, hipGetLastError());
    HCHECK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ------------------------------------------------------------------------------- forward
extern "C" int dne_set_members(dne_handle *h, int n, const int32_t *slot, const int64_t *off, const float *scale) {
    DeviceGuard dg(h);
    if (check_n(h, n)) return -1;
    for (int i = 0; i < n; i++) {
        if (slot[i] < 0 || slot[i] >= h->base_cap) return h->fail("member %d: base slot %d not allocated", i, slot[i]);
        if (check_noise_range(h, off[i], h->L.P)) return -1;
    }
    HCHECK(h, hipMemcpyAsync(h->m_slot, slot, n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HCHECK(h, hipMemcpyAsync(h->m_off, off, n * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HCHECK(h, hipMemcpyAsync(h->m_scale, scale, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HCHECK(h, hipStreamSynchronize(h->stream));
    h->uniform_base = true;
    for (int i = 0; i < n; i++) h->uniform_base = h->uniform_base && slot[i] == slot[0];
    h->dense_scale = 1.0;
    h->antithetic_slot0 = h->uniform_base && slot[0] == 0 && n % 2 == 0;
    for (int i = 0; i + 1 < n && h->antithetic_slot0; i += 2)
        h->antithetic_slot0 = off[i] == off[i + 1] && scale[i] == -scale[i + 1];
    h->pair_sigma = n > 0 ? scale[0] : 0.0f;
    h->pair_sigma_uniform = h->antithetic_slot0;
    for (int i = 0; i < n && h->pair_sigma_uniform; i += 2) h->pair_sigma_uniform = scale[i] == scale[0];
    h->members_materialized = false;
    h->host_slot.assign(slot, slot + n); h->host_off.assign(off, off + n); h->host_scale.assign(scale, scale + n);
    return 0;
}

// policies.py:399: the reference batch through every member's perturbed network -> per-member BN scale/shift
static int ref_pass(dne_handle *h, int n) {
    if (h->L.kind != DNE_KIND_ES) return 0;
    if (!h->ref_set) return h->fail("reference batch not set (dne_set_ref_batch)");
    const int F = h->F;
    const FwdArgs A = h->fwd(false);
    // chunks alternate between two streams with their own scratch: the statistics kernels (one workgroup per
    // member, latency-bound) of one chunk run under the MFMA convolutions of the next
    const int nways = n > h->ref_chunk ? 2 : 1;
    if (nways > 1) {
        HCHECK(h, hipEventRecord(h->ev_ref[0], h->stream));
        HCHECK(h, hipStreamWaitEvent(h->sub_streams[1], h->ev_ref[0], 0));
    }
    int c = 0;
    for (int m0 = 0; m0 < n; m0 += h->ref_chunk, c++) {
        const int nc = std::min(h->ref_chunk, n - m0), w = c % nways;
        hipStream_t st = h->sub_streams[w];
        float *y1 = h->y1r[w], *y2 = h->y2r[w], *y3p = h->y3pr[w];
        float *fr1 = h->fr1[w], *fr2 = h->fr2[w];
        const int fpw = h->conv1_fpw >= 8 ? 8 : h->conv1_fpw >= 4 ? 4 : h->conv1_fpw >= 2 ? 2 : 1;   // F is a multiple of 8
#define C1R(FPW) hipLaunchKernelGGL(k_conv1_ref<FPW>, dim3(nc * F / FPW), dim3(256), 0, st, A, F, m0, (const uint8_t *)h->ref, y1, fr1)
#define C1S(FPW) hipLaunchKernelGGL(k_conv1_ref_shared<FPW>, dim3((nc + 7) / 8 * (F / FPW)), dim3(512), RF_FRAME * sizeof(float), st, A, F, m0, nc, (const float *)h->ref_f32, y1, fr1)
        if (h->conv1_shared && fpw == 8 && F % 16 == 0) C1S(16);
        else if (h->conv1_shared && fpw == 8) C1S(8);
        else if (fpw == 8) C1R(8); else if (fpw == 4) C1R(4); else if (fpw == 2) C1R(2); else C1R(1);
#undef C1S
#undef C1R
        // the convolutions leave per-frame moments behind; scale / shift per member is a 128-term sum per channel
        hipLaunchKernelGGL((k_bn_finalize<16>), dim3((nc * 16 + 255) / 256), dim3(256), 0, st, A, m0, nc, F, (const float *)fr1, 441, 0,
                           h->L.c1b, h->L.bn1b, h->L.bn1g);
        const bool mc_fc = F == 16 || F == 32 || F == 64 || F == 128;   // fc on the matrix cores: y2 rows padded to 128 positions
        if (mc_fc)   // 16 frames per workgroup (32 / 64 measured the same: 11.46 / 11.42 / 11.50 ms per 5000 members)
            hipLaunchKernelGGL((k_conv2_ref<16, true>), dim3(nc * (F / 16)), dim3(256), 0, st, A, F, m0, (const float *)y1, y2, fr2);
        else if (F % 8 == 0 && h->conv2_ref_fpw == 8)
            hipLaunchKernelGGL((k_conv2_ref<8, false>), dim3(nc * (F / 8)), dim3(256), 0, st, A, F, m0, (const float *)y1, y2, fr2);
        else if (F % 4 == 0 && h->conv2_ref_fpw >= 4)
            hipLaunchKernelGGL((k_conv2_ref<4, false>), dim3(nc * (F / 4)), dim3(256), 0, st, A, F, m0, (const float *)y1, y2, fr2);
        else
            hipLaunchKernelGGL((k_conv2<true>), dim3(nc * F), dim3(256), 0, st, A, (const int *)nullptr, 1, F, m0,
                               (const float *)y1, y2, 1, fr2);
        hipLaunchKernelGGL((k_bn_finalize<32>), dim3((nc * 32 + 255) / 256), dim3(256), 0, st, A, m0, nc, F, (const float *)fr2, 121, 32,
                           h->L.c2b, h->L.bn2b, h->L.bn2g);
        if (mc_fc) {   // matrix-core path
            const int grid = (nc + 7) / 8 * 8 * 4;   // (member, quarter) workgroups, those of a member on one XCD
#define FCREF(MT, NG) hipLaunchKernelGGL((k_fc_ref<MT>), dim3(grid * NG), dim3(256), MT == 4 ? FCREF4_LDS : 0, st, A, nc, m0, F, (const float *)y2, y3p)
            if (F == 16) FCREF(1, 1); else if (F == 32) FCREF(2, 1); else if (F == 64) FCREF(4, 1); else FCREF(4, 2);   // 128 frames: two groups of 64
#undef FCREF
            hipLaunchKernelGGL(k_bn3_partials, dim3(nc), dim3(256), 0, st, A, m0, F, (const float *)y3p);
        } else {
            const int nfg = F / 8;   // generic path: y3 rows live in the partial buffer of this way
            hipLaunchKernelGGL((k_fc<8, true, true, 4>), dim3((nc + 7) / 8 * 8 * nfg), dim3(256), 0, st, A,
                               (const int *)nullptr, nc, F, m0, (const float *)y2, y3p, (int32_t *)nullptr,
                               (float *)nullptr);
            hipLaunchKernelGGL(k_bn3_rows, dim3(nc), dim3(256), 0, st, A, m0, F, (const float *)y3p);
        }
    }
    if (nways > 1) {
        HCHECK(h, hipEventRecord(h->ev_ref[1], h->sub_streams[1]));
        HCHECK(h, hipStreamWaitEvent(h->stream, h->ev_ref[1], 0));
    }
    HCHECK(h, hipGetLastError());
    return 0;
}

// profiling build (-DDNE_PHASE_CLOCK): the milestones of the last launches, [6 kernels][128 workgroups][8]; -1 in the product library
extern "C" int dne_debug_phase_clock(dne_handle *h, long long *out) {
#ifdef DNE_PHASE_CLOCK
    HCHECK(h, hipDeviceSynchronize());
    HCHECK(h, hipMemcpyFromSymbol(out, HIP_SYMBOL(dne::g_phase), sizeof(long long) * 6 * 128 * 8));
    return 0;
#else
    (void)out;
    return h->fail("dne_debug_phase_clock: this library was built without DNE_PHASE_CLOCK (make clock)");
#endif
}

// profiling build: the per-workgroup records of the lock-step's kernels (env_synth.h: g_wgclk), oldest overwritten; returns the number
// of records ever written through *count (the ring holds the last min(count, cap)); reset = 1 clears the cursor afterwards
extern "C" int dne_debug_wg_clock(dne_handle *h, long long *out, unsigned cap, unsigned *count, int reset) {
#ifdef DNE_PHASE_CLOCK
    HCHECK(h, hipDeviceSynchronize());
    unsigned n = 0;
    HCHECK(h, hipMemcpyFromSymbol(&n, HIP_SYMBOL(dne::g_wgclk_n), sizeof(n)));
    if (count) *count = n;
    if (out) HCHECK(h, hipMemcpyFromSymbol(out, HIP_SYMBOL(dne::g_wgclk), sizeof(long long) * 4 * std::min(cap, dne::WGCLK_CAP)));
    if (reset) { n = 0; HCHECK(h, hipMemcpyToSymbol(HIP_SYMBOL(dne::g_wgclk_n), &n, sizeof(n))); }
    return 0;
#else
    (void)out; (void)cap; (void)count; (void)reset;
    return h->fail("dne_debug_wg_clock: this library was built without DNE_PHASE_CLOCK (make clock)");
#endif
}

// profiling build: k_fc_duo's tick stamps of its last launch (env_synth.h: g_duo_tick [64][8][288][2], g_duo_plan [64][8][8])
extern "C" int dne_debug_duo_ticks(dne_handle *h, long long *ticks, long long *plan) {
#ifdef DNE_PHASE_CLOCK
    HCHECK(h, hipDeviceSynchronize());
    HCHECK(h, hipMemcpyFromSymbol(ticks, HIP_SYMBOL(dne::g_duo_tick), sizeof(long long) * dne::DUO_TICK_WGS * 8 * dne::DUO_TICK_MAX * 2));
    HCHECK(h, hipMemcpyFromSymbol(plan, HIP_SYMBOL(dne::g_duo_plan), sizeof(long long) * dne::DUO_TICK_WGS * 8 * 8));
    return 0;
#else
    (void)ticks; (void)plan;
    return h->fail("dne_debug_duo_ticks: this library was built without DNE_PHASE_CLOCK (make clock)");
#endif
}

extern "C" int dne_ref_pass(dne_handle *h, int n) {
    DeviceGuard dg(h);
    if (check_n(h, n)) return -1;
    if (h->L.kind != DNE_KIND_ES) return h->fail("dne_ref_pass: GAAtariPolicy has no reference batch");
    if (ref_pass(h, n)) return -1;
    HCHECK(h, hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int dne_get_bn(dne_handle *h, int n, float *out) {
    DeviceGuard dg(h);
    if (check_n(h, n)) return -1;
    HCHECK(h, hipStreamSynchronize(h->stream));
    HCHECK(h, hipMemcpy(out, h->bn, (size_t)n * 608 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// the batch moments of the last reference pass (tf.contrib.layers.batch_norm with decay = 0 leaves exactly these in
// moving_mean / moving_variance, policies.py:322-328): [n][608] = mean1[16] var1[16] mean2[32] var2[32] mean3[256] var3[256]
extern "C" int dne_get_bn_moments(dne_handle *h, int n, float *out) {
    DeviceGuard dg(h);
    if (check_n(h, n)) return -1;
    if (h->L.kind != DNE_KIND_ES) return h->fail("dne_get_bn_moments: GAAtariPolicy has no batch norm");
    HCHECK(h, hipStreamSynchronize(h->stream));
    HCHECK(h, hipMemcpy(out, h->bn_mom, (size_t)n * 608 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// one policy decision for the groups in `list` (count groups of gsize members)
static void launch_forward(dne_handle *h, const int *list, int count, int gsize, bool use_done, hipStream_t st = nullptr,
                           bool act2 = false /* leave relu(bn2(y2)) instead of y2: the window's fc is k_fc_ring */) {
    if (!st) st = h->stream;
    const FwdArgs A = h->fwd(use_done);
    if (h->large) {   // LargeModel: three matrix-core convolutions (forward_large.h); members are single (GA)
        constexpr size_t l2 = lconv_mfma_lds_bytes<32, 4, 2, 11, 34>(), l3 = lconv_mfma_lds_bytes<64, 3, 1, 11, 68>();
        hipLaunchKernelGGL(k_lconv1, dim3(count * 2), dim3(256), 0, st, A, list, (const uint8_t *)h->stacks, h->y1);
        const int ns = count <= 128 ? 4 : count <= 256 ? 2 : 1;   // few members: one workgroup per 16-channel tile
        if (h->members_materialized) {
            hipLaunchKernelGGL((k_lconv_mfma<32, 64, 4, 2, 21, 11, 1, 34, false>), dim3(count * ns), dim3(256), l2, st, A, list, A.L.c2w, A.L.c2b, (const float *)h->y1, h->y2, ns);
            hipLaunchKernelGGL((k_lconv_mfma<64, 64, 3, 1, 11, 11, 1, 68, false>), dim3(count * ns), dim3(256), l3, st, A, list, A.L.c3w, A.L.c3b, (const float *)h->y2, h->y3, ns);
        } else {
            hipLaunchKernelGGL((k_lconv_mfma<32, 64, 4, 2, 21, 11, 1, 34, true>), dim3(count * ns), dim3(256), l2, st, A, list, A.L.c2w, A.L.c2b, (const float *)h->y1, h->y2, ns);
            hipLaunchKernelGGL((k_lconv_mfma<64, 64, 3, 1, 11, 11, 1, 68, true>), dim3(count * ns), dim3(256), l3, st, A, list, A.L.c3w, A.L.c3b, (const float *)h->y2, h->y3, ns);
        }
        return;
    }
    const bool es = h->L.kind == DNE_KIND_ES;
    const int items = count * gsize;
    // few members left: several workgroups per member (conv1: 28 position tiles over 4 or 7 workgroups; conv2: 8 over 2 or 4)
    const int s1 = items <= h->conv_split_max ? 7 : items <= h->conv_split_mid ? 4 : 1, s2 = items <= h->conv_split_max ? 4 : items <= 2 * h->conv_split_mid ? 2 : 1;
    if (h->conv_fused && items >= h->conv_fused_min && !h->dbg_skip) {   // one workgroup per member through both convolutions, y1 stays in LDS
        float *y1 = use_done ? nullptr : h->y1;                           // dne_act / debug_activations want y1; evaluations do not
        if (es) hipLaunchKernelGGL((k_conv12<true>), dim3(items), dim3(256), sizeof(Conv12Lds), st, A, list, gsize, (const uint8_t *)h->stacks, y1, h->y2, act2 ? 1 : 0);
        else hipLaunchKernelGGL((k_conv12<false>), dim3(items), dim3(256), sizeof(Conv12Lds), st, A, list, gsize, (const uint8_t *)h->stacks, y1, h->y2, 0);
        return;
    }
    if (items <= h->conv12t_max && !h->dbg_skip) {   // the tail: four workgroups per member through both convolutions, no y1 round trip
        float *y1 = use_done ? nullptr : h->y1;
        if (es) hipLaunchKernelGGL((k_conv12t<true>), dim3(items * 4), dim3(512), sizeof(Conv12Lds), st, A, list, gsize, (const uint8_t *)h->stacks, y1, h->y2);
        else hipLaunchKernelGGL((k_conv12t<false>), dim3(items * 4), dim3(512), sizeof(Conv12Lds), st, A, list, gsize, (const uint8_t *)h->stacks, y1, h->y2);
        if (act2) hipLaunchKernelGGL(k_y2_activate, dim3(items), dim3(256), 0, st, A, list, gsize, h->y2);
        return;
    }
    if (!(h->dbg_skip & 1))
    hipLaunchKernelGGL(k_conv1, dim3(items * s1), dim3(256), 0, st, A, list, gsize, 1, 0,
                       (const uint8_t *)h->stacks, (const uint8_t *)nullptr, h->y1, s1);
    if (h->dbg_skip & 2) return;
    if (es) hipLaunchKernelGGL((k_conv2<true>), dim3(items * s2), dim3(256), 0, st, A, list, gsize, 1, 0, (const float *)h->y1, h->y2, s2, (float *)nullptr);
    else hipLaunchKernelGGL((k_conv2<false>), dim3(items * s2), dim3(256), 0, st, A, list, gsize, 1, 0, (const float *)h->y1, h->y2, s2, (float *)nullptr);
    if (act2) hipLaunchKernelGGL(k_y2_activate, dim3(items), dim3(256), 0, st, A, list, gsize, h->y2);
}

static void launch_fc(dne_handle *h, const int *list, int count, int gsize, float *logits, hipStream_t st = nullptr,
                      bool out_fused = false /* tail only: the caller runs k_tail_step instead of k_out */,
                      const int *order = nullptr /* the window's units in noise-table order (k_unit_order), duo regime only */,
                      hipEvent_t after_stream_kernel = nullptr /* duo regime: recorded between k_fc_duo and k_out (profiling) */) {
    if (!st) st = h->stream;
    // inside an evaluation (no logits requested) groups whose members are all done are skipped: they stay in the list until
    // the next compaction, and streaming their weights would be wasted bandwidth
    const FwdArgs A = h->fwd(logits == nullptr);
    const bool es = h->L.kind == DNE_KIND_ES;
    if (h->large) {   // LargeModel: streamed 7744 x 512 fc (two 256-column halves per member), then relu + output layer + argmax
        const dim3 lg(std::min(2 * count, 2 * h->fc_grid));
        if (count <= h->lfc_cols_max) {   // few members: eight workgroups each
            if (h->members_materialized) hipLaunchKernelGGL((k_lfc_cols<false>), dim3(8 * count), dim3(256), 0, st, A, list, (const float *)h->y3, h->y3t);
            else hipLaunchKernelGGL((k_lfc_cols<true>), dim3(8 * count), dim3(256), 0, st, A, list, (const float *)h->y3, h->y3t);
        } else if (h->members_materialized) {
            if (h->fc_rb == 8 && h->lfc_pad == 1) hipLaunchKernelGGL((k_lfc<false, 8, 1>), lg, dim3(256), 0, st, A, list, 2 * count, (const float *)h->y3, h->y3t);
            else if (h->fc_rb == 8 && h->lfc_pad == 2) hipLaunchKernelGGL((k_lfc<false, 8, 2>), lg, dim3(256), 0, st, A, list, 2 * count, (const float *)h->y3, h->y3t);
            else if (h->fc_rb == 8) hipLaunchKernelGGL((k_lfc<false, 8>), lg, dim3(256), 0, st, A, list, 2 * count, (const float *)h->y3, h->y3t);
            else hipLaunchKernelGGL((k_lfc<false, 4>), lg, dim3(256), 0, st, A, list, 2 * count, (const float *)h->y3, h->y3t);
        } else hipLaunchKernelGGL((k_lfc<true, 4>), lg, dim3(256), 0, st, A, list, 2 * count, (const float *)h->y3, h->y3t);
        hipLaunchKernelGGL(k_lout, dim3(count), dim3(256), 0, st, A, list, (const float *)h->y3t, h->action, logits);
        return;
    }
    if (h->sub_now && !logits && h->y3s) {   // mid range: one wave per sub-slice chain, folded by the head
        const int total_waves_at_1 = 32 * count;
        // sub-slices per wave: about 8000 waves (one round of the whole machine) whatever the width
        const int spw = h->fc_sub_spw ? h->fc_sub_spw : count * h->fc_sub_nsub <= 320 ? 1 : count * h->fc_sub_nsub <= 640 ? 2 : count * h->fc_sub_nsub <= 1280 ? 4 : 8;
        const int waves = total_waves_at_1 / spw, blocks = std::min((waves + 3) / 4, h->fc_sub_grid);
        // eval_core's sub_regime admits exactly two populations: ES pairs and GA children written out
        if (es) hipLaunchKernelGGL((k_fc_sub<2, true, true>), dim3(blocks), dim3(256), 0, st, A, list, count, spw, h->fc_sub_prio, (const float *)h->y2, h->y3s);
        else hipLaunchKernelGGL((k_fc_sub<1, false, false>), dim3(blocks), dim3(256), 0, st, A, list, count, spw, h->fc_sub_prio, (const float *)h->y2, h->y3s);
        if (out_fused) return;   // the caller runs k_tail_step (FwdArgs::sub_sums tells it to fold the chain sums)
        if (es) hipLaunchKernelGGL((k_out<2, true>), dim3(count), dim3(256), 0, st, A, list, (const float *)h->y3s, h->y3, h->action, (float *)nullptr);
        else hipLaunchKernelGGL((k_out<1, false>), dim3(count), dim3(256), 0, st, A, list, (const float *)h->y3s, h->y3, h->action, (float *)nullptr);
        return;
    }
    if (count <= h->fc_tail_max) {   // latency-bound regime: 4 workgroups per group + a separate output-layer kernel
#define FCT(NV, BN)                                                                                                          \
    do {                                                                                                                     \
        if (NV == 1 && !BN && h->members_materialized) {   /* GA children written out: plain rows, no noise stream */        \
            if (count <= h->fc_quad_max) hipLaunchKernelGGL((k_fc_quad<1, false, false>), dim3(count * 64), dim3(256), 0, st, A, list, (const float *)h->y2, h->y3t); \
            else hipLaunchKernelGGL((k_fc_tail<1, false, false>), dim3(count * 16), dim3(512), 0, st, A, list, (const float *)h->y2, h->y3t); \
        } else if (count <= h->fc_quad_max) hipLaunchKernelGGL((k_fc_quad<NV, BN>), dim3(count * 64), dim3(256), 0, st, A, list, (const float *)h->y2, h->y3t); \
        else if (count <= h->fc_tailk_max) hipLaunchKernelGGL((k_fc_tail<NV, BN>), dim3(count * 16), dim3(512), 0, st, A, list, (const float *)h->y2, h->y3t); \
        else hipLaunchKernelGGL((k_fc_cols<NV, BN>), dim3(count * 4), dim3(256), 0, st, A, list, (const float *)h->y2, h->y3t); \
        if (!out_fused) hipLaunchKernelGGL((k_out<NV, BN>), dim3(count), dim3(256), 0, st, A, list, (const float *)h->y3t, h->y3, h->action, logits); \
    } while (0)
        if (gsize == 2) { if (es) FCT(2, true); else FCT(2, false); }
        else { if (es) FCT(1, true); else FCT(1, false); }
#undef FCT
        return;
    }
    if (h->duo_now && !logits && order && gsize == 2 && es) {   // table-ordered units: adjacent (pair, k-slice) units share their noise rows
        const bool solo = h->duo_solo_now;
        const bool sweep = h->duo_sweep && (!solo || h->duo_sweep > 1);
        const int duo_grid = h->duo_grid ? h->duo_grid : h->fc_grid;
        const int n_units = 4 * count, items = ((solo ? n_units : (n_units + 1) / 2) + 3) / 4, blocks = std::min(items, duo_grid);
        const size_t out_lds = (size_t)h->out_lds_kb * 1024;   // an LDS reservation nobody uses: it only bounds k_out's workgroups per CU next to the streaming fc
        const int flags = h->duo_lag | (solo ? 256 : 0) | (h->fc_prio << 9) | ((h->duo_sync - 1) << 11);
        if (h->ring_now) {   // one unit per wave, eight units per workgroup whatever the regime
            const int ring_blocks = std::min((n_units + 7) / 8, duo_grid);
            if (h->ring_pre_now) hipLaunchKernelGGL((k_fc_ring<true, 8>), dim3(ring_blocks), dim3(576), 0, st, A, order, n_units, (const float *)h->y2, h->y3t, (const float *)h->theta_perm, h->fc_prio << 9, (const float *)h->noise_pre);
            else hipLaunchKernelGGL((k_fc_ring<false, 8>), dim3(ring_blocks), dim3(576), 0, st, A, order, n_units, (const float *)h->y2, h->y3t, (const float *)h->theta_perm, h->fc_prio << 9, (const float *)A.noise);
        }
        else if (sweep && h->duo_fat) hipLaunchKernelGGL((k_fc_duo<2, true, true, 8, true>), dim3(blocks), dim3(256), 0, st, A, order, n_units, (const float *)h->y2, h->y3t, flags);
        else if (sweep) hipLaunchKernelGGL((k_fc_duo<2, true, true>), dim3(blocks), dim3(256), 0, st, A, order, n_units, (const float *)h->y2, h->y3t, flags);
        else hipLaunchKernelGGL((k_fc_duo<2, true>), dim3(blocks), dim3(256), 0, st, A, order, n_units, (const float *)h->y2, h->y3t, flags);
        if (after_stream_kernel) hipEventRecord(after_stream_kernel, st);
        if (out_fused) return;   // the caller runs k_tail_step: policy head + emulator step in one launch
        hipLaunchKernelGGL((k_out<2, true>), dim3(count), dim3(256), out_lds, st, A, list, (const float *)h->y3t, h->y3, h->action, (float *)nullptr);
        return;
    }
    if (gsize == 2 && es && h->uniform_base && h->fc2_now && !logits) {   // two pairs per work item share the base rows
        const int items = (count + 1) / 2, blocks = std::min(items, h->fc_grid);
        if (h->fc_rb == 2) hipLaunchKernelGGL((k_fc2<true, 2>), dim3(blocks), dim3(256), 0, st, A, list, count, (const float *)h->y2, h->y3, h->action);
        else hipLaunchKernelGGL((k_fc2<true, 4>), dim3(blocks), dim3(256), 0, st, A, list, count, (const float *)h->y2, h->y3, h->action);
        return;
    }
    const int fc_blocks = std::min(count, h->fc_grid);   // persistent grid (an even groups-per-block split measured slower)
#define FC(NV, BN, RB) hipLaunchKernelGGL((k_fc<NV, false, BN, RB>), dim3(fc_blocks), dim3(256), 0, st, A, list, count, 1, 0, (const float *)h->y2, h->y3, h->action, logits)
#define FCR(NV, BN) do { if (h->fc_rb == 2) FC(NV, BN, 2); else if (h->fc_rb == 8) FC(NV, BN, 8); else FC(NV, BN, 4); } while (0)
    if (gsize == 1 && !es && h->members_materialized) {   // GA children written out once per generation: plain rows, no noise stream
        if (h->fc_rb == 8) hipLaunchKernelGGL((k_fc<1, false, false, 8, false>), dim3(fc_blocks), dim3(256), 0, st, A, list, count, 1, 0, (const float *)h->y2, h->y3, h->action, logits);
        else hipLaunchKernelGGL((k_fc<1, false, false, 4, false>), dim3(fc_blocks), dim3(256), 0, st, A, list, count, 1, 0, (const float *)h->y2, h->y3, h->action, logits);
    } else if (gsize == 2) { if (es) FCR(2, true); else FCR(2, false); }
    else { if (es) FCR(1, true); else FCR(1, false); }
#undef FCR
#undef FC
}

extern "C" int dne_act(dne_handle *h, int n, int32_t *actions, float *logits) {
    DeviceGuard dg(h);
    if (check_n(h, n)) return -1;
    launch_forward(h, nullptr, n, 1, false);
    launch_fc(h, nullptr, n, 1, h->logits);
    HCHECK(h, hipGetLastError());
    HCHECK(h, hipStreamSynchronize(h->stream));
    if (actions) HCHECK(h, hipMemcpy(actions, h->action, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (logits) HCHECK(h, hipMemcpy(logits, h->logits, (size_t)n * h->cfg.n_actions * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int dne_debug_activations(dne_handle *h, int member, float *y1, float *y2, float *y3) {
    DeviceGuard dg(h);
    if (member < 0 || member >= h->M) return h->fail("bad member");
    if (h->large) return h->fail("dne_debug_activations: LargeModel engines use dne_debug_activations_large");
    HCHECK(h, hipStreamSynchronize(h->stream));
    if (y1) HCHECK(h, hipMemcpy(y1, h->y1 + (size_t)member * 7056, 7056 * sizeof(float), hipMemcpyDeviceToHost));
    if (y2) HCHECK(h, hipMemcpy(y2, h->y2 + (size_t)member * 3872, 3872 * sizeof(float), hipMemcpyDeviceToHost));
    if (y3) HCHECK(h, hipMemcpy(y3, h->y3 + (size_t)member * 256, 256 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// LargeModel: raw (bias added, pre-relu) outputs of conv1 [441][32], conv2 / conv3 [121][64] and the fc [512] after dne_act
extern "C" int dne_debug_activations_large(dne_handle *h, int member, float *y1, float *y2, float *y3, float *y4) {
    DeviceGuard dg(h);
    if (member < 0 || member >= h->M) return h->fail("bad member");
    if (!h->large) return h->fail("dne_debug_activations_large needs a DNE_KIND_GA_LARGE engine");
    HCHECK(h, hipStreamSynchronize(h->stream));
    if (y1) HCHECK(h, hipMemcpy(y1, h->y1 + (size_t)member * 14112, 14112 * sizeof(float), hipMemcpyDeviceToHost));
    if (y2) HCHECK(h, hipMemcpy(y2, h->y2 + (size_t)member * 7744, 7744 * sizeof(float), hipMemcpyDeviceToHost));
    if (y3) HCHECK(h, hipMemcpy(y3, h->y3 + (size_t)member * 7744, 7744 * sizeof(float), hipMemcpyDeviceToHost));
    if (y4) HCHECK(h, hipMemcpy(y4, h->y3t + (size_t)member * 512, 512 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------------------------- batch evaluation
// policies.py:378-429 / 473-513 for n members at once: reset, (ES) reference pass, then lock-step
// act -> env.step over the shrinking list of active groups until every episode is done.
// The active groups are split into `nsub` independent sub-batches, each stepped on its own HIP stream: while
// one sub-batch streams its noise slices through the HBM-bound fc kernel, another runs its MFMA convolutions
// and its emulator frames, so the memory system and the matrix/vector pipes are busy at the same time.
static int eval_core(dne_handle *h, int n, int gsize, int tslimit, const uint32_t *env_seed, float *returns,
                     float *signreturns, int32_t *lengths, uint8_t *bc_out) {
    h->tt_on = false;
    // per-evaluation launch state (tail table in the kernel arguments, the regime flags) never outlives this call, whichever
    // return is taken: a later dne_act / dne_debug_activations must not decode members from a stale table
    struct ClearOnExit { dne_handle *h; ~ClearOnExit() { h->tt_on = false; h->sub_now = false; h->ring_now = false; } } clear_on_exit{h};
    if (n % gsize) return h->fail("member count %d not a multiple of the group size %d", n, gsize);
    if (tslimit <= 0) return h->fail("timestep limit must be positive");
    if (bc_out && !h->bc) return h->fail("behaviour characterisations requested but the engine was created with record_bc = 0");
    const bool prof = h->cfg.profile_events != 0;
    // record_bc engines always record (the trajectories feed dne_novelty_batch on the device); bc_out only controls the download
    const int bc_mode = h->bc ? (h->L.kind == DNE_KIND_ES && !h->cfg.bc_final_only ? 1 : 2) : 0;
    // rows past an episode's length are never read on the device (dne_novelty_batch takes the lengths); zero them only
    // when the whole buffer is about to be downloaded
    if (bc_mode == 1 && bc_out) HCHECK(h, hipMemsetAsync(h->bc, 0, (size_t)n * h->cfg.bc_max_steps * 128, h->stream));
    if (bc_mode == 2) HCHECK(h, hipMemsetAsync(h->bc, 0, (size_t)n * 128, h->stream));
    HCHECK(h, hipEventRecord(h->ev_a, h->stream));
    HCHECK(h, hipMemcpyAsync(h->seeds, env_seed, n * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    launch_env_reset(h, n);
    if (prof) HCHECK(h, hipEventRecord(h->event(0), h->stream));
    h->trace("eval: %d members (groups of %d), tslimit %d: reset launched", n, gsize, tslimit);
    if (h->debug_sync) { HCHECK(h, hipStreamSynchronize(h->stream)); HCHECK(h, hipGetLastError()); }
    if (h->theta_perm && h->antithetic_slot0 && gsize == 2)   // k_fc_ring reads base slot 0's fc matrix in its own column order
        hipLaunchKernelGGL(k_theta_perm, dim3(3872 + 16), dim3(256), 0, h->stream, (const float *)(h->bases + h->L.fcw), h->theta_perm, 3872);
    if (ref_pass(h, n)) return -1;
    if (h->debug_sync) { HCHECK(h, hipDeviceSynchronize()); HCHECK(h, hipGetLastError()); }
    h->trace("eval: reference pass %s", h->debug_sync ? "done" : "launched");
    HCHECK(h, hipEventRecord(h->event(1), h->stream));
    const int groups = n / gsize;
    hipLaunchKernelGGL(k_iota, dim3((groups + 255) / 256), dim3(256), 0, h->stream, h->list_a, groups);
    HCHECK(h, hipStreamSynchronize(h->stream));

    // One global list of active groups, compacted every burst of 32 lock-steps (16 in the tail).  Within a burst the list is cut
    // into nsub equal windows, each stepped on its own stream, so that while one window streams its noise slices
    // (HBM) the others run their MFMA convolutions and emulator frames.  nsub follows the active count:
    // two free-running streams at full width, three in the mid range (where no single kernel fills the chip),
    // one when only a handful of episodes are left (measured: tools/kbench.py sweeps).
    // the sub-slice fc's range: GA children written out (plain rows), optionally ES pairs (DNE_FC_SUB=2); never the LargeModel
    auto sub_regime = [&](int total) {
        if (h->large || !h->y3s || total < h->fc_sub_min || total > h->fc_sub_max) return false;
        if (h->L.kind == DNE_KIND_ES) return h->fc_sub >= 2 && gsize == 2 && total < h->fc_duo_min;
        return h->fc_sub >= 1 && gsize == 1 && h->members_materialized;
    };
    auto pick_nsub = [&](int total) {
        // measured (tools/kbench.py sweeps): k_fc2 wants 3 windows at full width and 4 in the upper mid range; below
        // ~400 groups the windows are sized to fit the column-split tail kernels (<= fc_tail_max groups each)
        int k = total >= 1900 ? h->nsub_full : total >= h->fc2_min_total ? h->nsub_mid : total > 4 * h->fc_tail_max ? 3
              : total >= 48 ? std::max(2, (total + h->fc_tail_max - 1) / h->fc_tail_max) : 1;
        if (sub_regime(total)) k = h->fc_sub_nsub;
        if (h->nsub_fixed > 0) k = h->nsub_fixed;
        k = std::min(k, (int)h->sub_streams.size());
        return std::max(1, std::min(k, total));
    };
    int *cur = h->list_a, *nxt = h->list_b;
    int total = groups;
    // tail table: with at most TT_MAX members left the window's member descriptors ride in the kernel arguments
    auto set_tail_table = [&](const int *host_list /* the active groups in list order, null = 0, 1, 2, ... */) {
        h->tt_on = false;
        if (!h->tt_enable || h->large || total * gsize > TT_MAX || (int)h->host_off.size() < n || pick_nsub(total) != 1) return;
        h->tt.n = total * gsize;
        for (int i = 0; i < total; i++)
            for (int v = 0; v < gsize; v++) {
                const int m = (host_list ? host_list[i] : i) * gsize + v, k = i * gsize + v;
                h->tt.member[k] = m; h->tt.slot[k] = h->host_slot[m]; h->tt.scale[k] = h->host_scale[m]; h->tt.off[k] = h->host_off[m];
            }
        h->tt_on = true;
    };
    set_tail_table(nullptr);
    EnvArgs E = h->env(bc_mode);
    int t = 0;
    long long group_steps = 0, launch_sets = 0;
    std::vector<std::array<size_t, 4>> evs;   // per profiled launch set: event indices {before, after conv, after fc, after env}
    size_t ne = 2;
    if (prof) {
        const size_t need = (size_t)h->sub_streams.size() * (size_t)tslimit + 64;
        if (need > h->launch_units_cap) {
            HCHECK(h, h->release(h->launch_units));
            HCHECK(h, h->alloc(&h->launch_units, need, "launch_units"));
            h->launch_units_cap = need;
        }
        HCHECK(h, hipMemsetAsync(h->launch_units, 0, need * sizeof(int32_t), h->stream));
        HCHECK(h, hipStreamSynchronize(h->stream));
    }
    hipEvent_t last_fc = nullptr;
    size_t fc_ring_pos = 0;
    // the profiled ("full") launches are one kernel: k_fc2 when this evaluation starts wide enough to use it, else k_fc
    const bool fc2_eval = h->fc_pairs == 2 && gsize == 2 && h->L.kind == DNE_KIND_ES && h->uniform_base && groups >= h->fc2_min_total;
    const bool duo_eval = !h->large && h->fc_duo && (h->L.kind == DNE_KIND_ES && gsize == 2) && groups >= h->fc_duo_min;
    // an evaluation that starts wide enough for k_fc_ring: its bracketed ("full") launches are that kernel's only -- one kernel per
    // roofline line; the k_fc_duo launches of its thinner lock-steps (DNE_FC_DUO_MIN .. DNE_DUO_SOLO_BELOW pairs) are not bracketed
    // the ring's range: pairs as dense in their stretch of the table as DNE_DUO_SOLO_BELOW (1500) pairs over the whole table, and enough of them to
    // fill the chip (DNE_RING_MIN) -- at one GPU "1500 of 2500 active", on a rank of two with its own half of the table "1000 of 1250"
    auto ring_dense = [&](int t) { return (double)t * h->dense_scale >= (double)h->duo_solo_below && t >= h->ring_min; };
    const bool ring_eval = duo_eval && h->L.kind == DNE_KIND_ES && gsize == 2 && h->theta_perm && h->antithetic_slot0 && h->duo_sweep &&
                           (h->ring_on > 1 || ring_dense(groups)) && (ring_dense(groups) || h->duo_sweep > 1);
    // the ring's DMA source: the table scaled by this evaluation's sigma (k_fc_ring<true>), made once per (table, sigma)
    h->ring_pre_now = false;
    if (ring_eval && h->ring_pre && h->pair_sigma_uniform) {
        if (!h->noise_pre && h->alloc(&h->noise_pre, h->noise_count + OVERFETCH_FLOATS, "noise_pre")) {
            h->trace("no room for the scaled table: the ring scales its rows itself");
            (void)hipGetLastError(); h->noise_pre = nullptr; h->ring_pre = 0;
        }
        if (h->noise_pre) {
            if (h->noise_pre_count != h->noise_count || h->noise_pre_sigma != h->pair_sigma) {
                static_assert(OVERFETCH_FLOATS % 4 == 0, "k_scale_table moves 16 bytes per step");
                const size_t n4 = (h->noise_count + OVERFETCH_FLOATS) / 4;   // (a table whose count is not a multiple of four leaves its last 1-3 padding floats unscaled: zeros either way)
                hipLaunchKernelGGL(k_scale_table, dim3(256 * 8), dim3(256), 0, h->stream, (const float *)h->noise, h->noise_pre, n4, h->pair_sigma);
                HCHECK(h, hipStreamSynchronize(h->stream));   // the windows' streams start behind the host
                h->noise_pre_count = h->noise_count; h->noise_pre_sigma = h->pair_sigma;
                h->trace("noise table scaled by %g for the ring", (double)h->pair_sigma);
            }
            h->ring_pre_now = true;
        }
    }
    while (total > 0 && t < tslimit) {
        const int burst = std::min(total <= h->fc_tail_max ? h->burst_tail : h->burst, tslimit - t);   // lock-steps until the next compaction
        const int nsub = pick_nsub(total);
        h->fc2_now = h->fc_pairs == 2 && total >= h->fc2_min_total;
        h->duo_now = !h->large && h->fc_duo && (h->L.kind == DNE_KIND_ES && gsize == 2) && total >= h->fc_duo_min &&
                     (size_t)4 * ((total + nsub - 1) / nsub) * sizeof(long long) <= 160 * 1024;   // k_unit_order ranks a window's keys in LDS
        h->duo_solo_now = total < h->duo_solo_below;
        h->sub_now = sub_regime(total);
        if (h->sub_now) h->duo_now = h->fc2_now = false;
        h->ring_now = h->duo_now && h->L.kind == DNE_KIND_ES && gsize == 2 && h->theta_perm && h->antithetic_slot0 &&
                      h->duo_sweep && (ring_dense(total) || h->duo_sweep > 1) && (h->ring_on > 1 || ring_dense(total));
        if (h->duo_now)   // the list only changes at a compaction: rank each window's units by table address once per burst
            for (int s = 0; s < nsub; s++) {
                const int lo = (int)((long long)total * s / nsub), cnt = (int)((long long)total * (s + 1) / nsub) - lo;
                if (cnt <= h->fc_tail_max) continue;
                hipLaunchKernelGGL(k_unit_order, dim3((4 * cnt + 255) / 256), dim3(256), (size_t)4 * cnt * sizeof(long long), h->sub_streams[s],
                                   (const int64_t *)h->m_off, (const int *)(cur + lo), cnt, gsize, h->unit_order + 4 * lo);
            }
        for (int st = 0; st < burst; st++) {
            for (int s = 0; s < nsub; s++) {
                const int lo = (int)((long long)total * s / nsub), cnt = (int)((long long)total * (s + 1) / nsub) - lo;
                if (cnt == 0) continue;
                hipStream_t sst = h->sub_streams[s];
                const int *lst = cur + lo;
                std::array<size_t, 4> e{};
                // events only around full-width launches: in the latency-bound tail every event packet is a bubble
                // (with k_fc2 enabled the profiled launches are exactly the k_fc2 ones: the roofline kernel of bench.py)
                const bool duo_win = h->duo_now && cnt > h->fc_tail_max;
                const bool pe = prof && (ring_eval ? duo_win && h->ring_now : duo_eval ? duo_win : fc2_eval ? h->fc2_now : cnt > h->fc_tail_max);
                if (pe) { e[0] = ne++; HCHECK(h, hipEventRecord(h->event(e[0]), sst)); }
                // fused policy head + emulator (+ render): while all windows together still fit the chip one workgroup per member
                const bool tail = !h->large && !h->sub_now && cnt <= h->fc_tail_max && total <= h->tail_fused_max;   // (the fused tail kernels are the small networks')
                // speculative tail: the emulator + renderer outcome of every action, inside the launches of this step's forward pass
                const bool spec = tail && nsub == 1 && h->spec_max > 0 && cnt * gsize <= h->spec_max &&
                                  cnt * gsize <= h->conv_split_max && !h->dbg_skip;
                if (spec) {
                    const int items = cnt * gsize, nact = h->cfg.n_actions, nb = h->spec_bands;
                    const FwdArgs A = h->fwd(true);
                    const bool es = h->L.kind == DNE_KIND_ES;
                    const int emu_blocks = (items * nact + 255) / 256;
                    if (st == 0 || !h->spec_conv1) {   // no conv1 candidates from the previous lock-step (a new burst = a new list)
                        hipLaunchKernelGGL(k_conv1_spec, dim3(items * 7 + emu_blocks), dim3(256), 0, sst, A, E, lst, gsize, h->y1, 7, items * 7, items, nact);
                        if (es) hipLaunchKernelGGL((k_conv2<true>), dim3(items * 4), dim3(256), 0, sst, A, lst, gsize, 1, 0, (const float *)h->y1, h->y2, 4, (float *)nullptr);
                        else hipLaunchKernelGGL((k_conv2<false>), dim3(items * 4), dim3(256), 0, sst, A, lst, gsize, 1, 0, (const float *)h->y1, h->y2, 4, (float *)nullptr);
                    } else if (es) {
                        hipLaunchKernelGGL((k_conv2_spec<true>), dim3(items * 4 + emu_blocks), dim3(256), 0, sst, A, E, lst, gsize, (const float *)h->y1, h->y2, 4, items * 4, items, nact, (const int32_t *)h->action);
                    } else {
                        hipLaunchKernelGGL((k_conv2_spec<false>), dim3(items * 4 + emu_blocks), dim3(256), 0, sst, A, E, lst, gsize, (const float *)h->y1, h->y2, 4, items * 4, items, nact, (const int32_t *)h->action);
                    }
#define FQS(NV, BN) hipLaunchKernelGGL((k_fc_quad_spec<NV, BN>), dim3(cnt * 64 + items * nact * nb), dim3(256), 0, sst, A, E, lst, gsize, (const float *)h->y2, h->y3t, cnt * 64, nact, nb)
                    if (gsize == 2) { if (es) FQS(2, true); else FQS(2, false); }
                    else { if (es) FQS(1, true); else FQS(1, false); }
#undef FQS
                    if (h->spec_conv1 && st + 1 < burst) {   // the choice + conv1 of every candidate (the last lock-step of a burst has no successor to use them)
                        if (es) hipLaunchKernelGGL((k_tail_select_conv1<true>), dim3(items + items * nact * 7), dim3(256), 0, sst, A, E, lst, gsize, tslimit, (const float *)h->y3t, h->y3, h->action, items, nact, 7);
                        else hipLaunchKernelGGL((k_tail_select_conv1<false>), dim3(items + items * nact * 7), dim3(256), 0, sst, A, E, lst, gsize, tslimit, (const float *)h->y3t, h->y3, h->action, items, nact, 7);
                    } else if (es) hipLaunchKernelGGL((k_tail_select<true>), dim3(items), dim3(256), 0, sst, A, E, lst, gsize, tslimit, (const float *)h->y3t, h->y3, h->action);
                    else hipLaunchKernelGGL((k_tail_select<false>), dim3(items), dim3(256), 0, sst, A, E, lst, gsize, tslimit, (const float *)h->y3t, h->y3, h->action);
                    if (h->debug_sync) {
                        hipError_t de = hipStreamSynchronize(sst);
                        if (de == hipSuccess) de = hipGetLastError();
                        if (de != hipSuccess) return h->fail("lock-step %d (speculative tail, %d active groups): %s", t + st, cnt, hipGetErrorString(de));
                    }
                    group_steps += cnt;
                    launch_sets++;
                    continue;
                }
                launch_forward(h, lst, cnt, gsize, true, sst, h->ring_now && duo_win);
                // optional: serialise the fc kernels of the windows (anti-phase); off by default, free-running measured faster
                const bool chain = nsub > 1 && cnt >= h->fc_chain_min;
                if (chain && last_fc) HCHECK(h, hipStreamWaitEvent(sst, last_fc, 0));
                if (pe) { e[1] = ne++; HCHECK(h, hipEventRecord(h->event(e[1]), sst)); }   // after the wait: brackets fc only
                if (pe) e[2] = ne++;
                const bool duo_head = (duo_win && h->duo_head_fused) || (h->sub_now && h->fc_sub_head);   // the fc leaves partial sums: head + emulator in one launch
                launch_fc(h, lst, cnt, gsize, nullptr, sst, tail || duo_head, h->duo_now ? h->unit_order + 4 * lo : nullptr,
                          pe && duo_win ? h->event(e[2]) : nullptr);   // duo: the bracket ends behind k_fc_duo, before k_out
                if (chain) { last_fc = h->fc_ring[fc_ring_pos++ % h->fc_ring.size()]; HCHECK(h, hipEventRecord(last_fc, sst)); }
                if (pe && !duo_win) HCHECK(h, hipEventRecord(h->event(e[2]), sst));
                E.step_counter = pe ? h->launch_units + evs.size() : nullptr;
                if (tail) {
                    const FwdArgs A = h->fwd(false);
                    const int items = cnt * gsize;
                    int nb = h->render_bands;
                    while (nb > 1 && items * nb > h->render_wg_max) nb /= 2;
#define TS(BN, R, THR) hipLaunchKernelGGL((k_tail_step<BN, R>), dim3(items), dim3(THR), 0, sst, A, E, lst, gsize, tslimit, (const float *)h->y3t, h->y3, h->action)
                    const bool es = h->L.kind == DNE_KIND_ES;
                    if (nb > 1) {   // few members left: policy head + emulator, then each frame over nb workgroups
                        if (es) TS(true, false, h->head_threads); else TS(false, false, h->head_threads);
                        hipLaunchKernelGGL(k_env_render, dim3(items * nb), dim3(h->band_threads), 0, sst, E, lst, gsize, 0, nb);
                    } else if (es) TS(true, true, 1024); else TS(false, true, 1024);
#undef TS
                } else if (duo_head) {
                    const FwdArgs A = h->fwd(false);
                    const int items = cnt * gsize;
                    const float *sums = h->sub_now ? h->y3s : h->y3t;
                    if (h->sub_render_fused && h->sub_now && !(h->dbg_skip & 4)) {   // head + emulator + renderer in one launch
                        if (h->L.kind == DNE_KIND_ES) hipLaunchKernelGGL((k_tail_step<true, true>), dim3(items), dim3(1024), 0, sst, A, E, lst, gsize, tslimit, sums, h->y3, h->action);
                        else hipLaunchKernelGGL((k_tail_step<false, true>), dim3(items), dim3(1024), 0, sst, A, E, lst, gsize, tslimit, sums, h->y3, h->action);
                    } else {
                    if (h->L.kind == DNE_KIND_ES) hipLaunchKernelGGL((k_tail_step<true, false>), dim3(items), dim3(h->head_threads), 0, sst, A, E, lst, gsize, tslimit, sums, h->y3, h->action);
                    else hipLaunchKernelGGL((k_tail_step<false, false>), dim3(items), dim3(h->head_threads), 0, sst, A, E, lst, gsize, tslimit, sums, h->y3, h->action);
                    if (!(h->dbg_skip & 4))
                        hipLaunchKernelGGL(k_env_render, dim3(items), dim3(items <= 192 ? 1024 : h->render_threads), 0, sst, E, lst, gsize, 0, 1);
                    }
                } else launch_env_step(h, E, lst, cnt, gsize, tslimit, sst);
                if (pe) { e[3] = ne++; HCHECK(h, hipEventRecord(h->event(e[3]), sst)); evs.push_back(e); }
                if (h->debug_sync) {
                    hipError_t de = hipStreamSynchronize(sst);
                    if (de == hipSuccess) de = hipGetLastError();
                    if (de != hipSuccess) return h->fail("lock-step %d window %d/%d (%d of %d active groups, %s fc): %s", t + st, s, nsub, cnt, total,
                                                         cnt <= h->fc_tail_max ? "tail" : h->fc2_now ? "k_fc2" : "k_fc", hipGetErrorString(de));
                }
                group_steps += cnt;
                launch_sets++;
            }
        }
        t += burst;
        for (int s = 1; s < nsub; s++) HCHECK(h, hipStreamSynchronize(h->sub_streams[s]));
        hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, h->stream, (const int32_t *)h->done, gsize, (const int *)cur,
                           total, nxt, h->count_dev);
        HCHECK(h, hipMemcpyAsync(h->count_host, h->count_dev, (8 + TT_MAX) * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HCHECK(h, hipStreamSynchronize(h->stream));
        total = *h->count_host;
        std::swap(cur, nxt);
        set_tail_table(h->count_host + 8);
        E.tt_n = h->tt_on ? h->tt.n : 0;
        if (h->tt_on) memcpy(E.tt_member, h->tt.member, sizeof(E.tt_member));
        h->trace("eval: lock-step %d, %d active groups", t, total);
    }
    h->tt_on = false;
    h->sub_now = false;
    HCHECK(h, hipGetLastError());
    HCHECK(h, hipEventRecord(h->ev_b, h->stream));
    HCHECK(h, hipMemcpyAsync(returns, h->ret, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (signreturns) HCHECK(h, hipMemcpyAsync(signreturns, h->sign, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HCHECK(h, hipMemcpyAsync(lengths, h->len, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HCHECK(h, hipStreamSynchronize(h->stream));
    if (bc_out && copy_d2h(h, bc_out, h->bc, bc_mode == 1 ? (size_t)n * h->cfg.bc_max_steps * 128 : (size_t)n * 128)) return -1;
    float ms = 0;
    HCHECK(h, hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
    dne_profile &P = h->prof;
    P.eval_ms = ms;
    P.fc_ms = P.conv_ms = P.env_ms = P.ref_ms = 0;
    P.fc_launches = launch_sets;
    P.fc_group_steps = group_steps;
    P.env_steps = 0;
    for (int i = 0; i < n; i++) P.env_steps += lengths[i];
    P.fc_full_ms = P.fc_full_launches = P.fc_full_units = 0;
    P.fc_full_kind = ring_eval ? 5 : duo_eval ? 3 : fc2_eval ? 2 : sub_regime(groups) ? 4 : 1;   // (an evaluation that STARTS in the sub-slice fc's range: its bracketed launches are k_fc_sub's)
    P.fc_full_union_ms = 0;
    if (prof) {
        HCHECK(h, hipEventElapsedTime(&ms, h->ev_pool[0], h->ev_pool[1]));
        P.ref_ms = ms;
        std::vector<int32_t> units(evs.size());
        if (!evs.empty()) HCHECK(h, hipMemcpy(units.data(), h->launch_units, evs.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        std::vector<std::pair<float, float>> iv;   // fc intervals relative to the start of the reference pass
        for (size_t i = 0; i < evs.size(); i++) {
            const auto &e = evs[i];
            HCHECK(h, hipEventElapsedTime(&ms, h->ev_pool[e[0]], h->ev_pool[e[1]])); P.conv_ms += ms;
            HCHECK(h, hipEventElapsedTime(&ms, h->ev_pool[e[1]], h->ev_pool[e[2]])); P.fc_ms += ms;
            P.fc_full_ms += ms; P.fc_full_launches += 1; P.fc_full_units += units[i];
            float t0 = 0;
            HCHECK(h, hipEventElapsedTime(&t0, h->ev_pool[0], h->ev_pool[e[1]]));
            iv.push_back({t0, t0 + ms});
            HCHECK(h, hipEventElapsedTime(&ms, h->ev_pool[e[2]], h->ev_pool[e[3]])); P.env_ms += ms;
        }
        std::sort(iv.begin(), iv.end());
        double uni = 0; float hi = -1;
        for (auto &p : iv) {   // length of the union of the intervals
            if (p.first > hi) { uni += p.second - p.first; hi = p.second; }
            else if (p.second > hi) { uni += p.second - hi; hi = p.second; }
        }
        P.fc_full_union_ms = uni;
    }
    return 0;
}

extern "C" int dne_es_eval(dne_handle *h, const int64_t *idx, int n, float sigma, int tslimit, const uint32_t *env_seed,
                           float *returns_n2, float *signreturns_n2, int32_t *lengths_n2, uint8_t *bc) {
    DeviceGuard dg(h);
    if (h->L.kind != DNE_KIND_ES) return h->fail("dne_es_eval needs an ESAtariPolicy engine");
    if (n <= 0 || 2 * n > h->M) return h->fail("%d pairs exceed max_members = %d", n, h->M);
    std::vector<int32_t> slot(2 * n, 0);
    std::vector<int64_t> off(2 * n);
    std::vector<float> sc(2 * n);
    for (int i = 0; i < n; i++) {
        off[2 * i] = off[2 * i + 1] = idx[i];
        sc[2 * i] = sigma;        // es.py:415 params + v
        sc[2 * i + 1] = -sigma;   // es.py:419 params - v
    }
    if (dne_set_members(h, 2 * n, slot.data(), off.data(), sc.data())) return -1;
    {   // how dense the pairs lie in the stretch of the table they cover (round 6: ranks that draw from their own stretch)
        int64_t lo = idx[0], hi = idx[0];
        for (int i = 1; i < n; i++) { lo = std::min(lo, idx[i]); hi = std::max(hi, idx[i]); }
        const double span = (double)(hi - lo) + (double)h->L.P;
        h->dense_scale = std::min(16.0, std::max(1.0, (double)h->noise_count / std::max(span, 1.0)));
    }
    return eval_core(h, 2 * n, 2, tslimit, env_seed, returns_n2, signreturns_n2, lengths_n2, bc);
}

extern "C" int dne_eval_members(dne_handle *h, int n, int tslimit, const uint32_t *env_seed, float *returns,
                                float *signreturns, int32_t *lengths, uint8_t *bc) {
    DeviceGuard dg(h);
    if (check_n(h, n)) return -1;
    return eval_core(h, n, 1, tslimit, env_seed, returns, signreturns, lengths, bc);
}

// ------------------------------------------------------------------------------- GA genomes
static void launch_normc(dne_handle *h, float *th) {
    const Layout &L = h->L;
    auto nc = [&](int off, int K, int C, float std) { hipLaunchKernelGGL(k_normc, dim3((C + 63) / 64), dim3(64), 0, h->stream, th + off, K, C, std); };
    auto z = [&](int off, int n) { hipLaunchKernelGGL(k_zero, dim3((n + 63) / 64), dim3(64), 0, h->stream, th + off, n); };
    nc(L.c1w, 256, 16, 1.0f); z(L.c1b, 16);
    nc(L.c2w, 256, 32, 1.0f); z(L.c2b, 32);
    nc(L.fcw, 3872, 256, 1.0f); z(L.fcb, 256);
    nc(L.ow, 256, L.nact, 0.1f); z(L.ob, L.nact);   // ac_init_std = 0.1, policies.py:434
}

// build theta(chain) into `slot`; `src_slot` >= 0 means chain[:src_len] is already materialised there
static int build_chain(dne_handle *h, int slot, const int64_t *seeds, const float *powers /*per seed or null*/, int nseeds, float sigma,
                       int src_slot, int src_len) {
    const int P = h->L.P, nb = (P + 255) / 256;
    float *dst = h->bases + (size_t)slot * h->base_stride;
    for (int s = 0; s < nseeds; s++)
        if (check_noise_range(h, seeds[s], P)) return -1;
    int start = 1;
    const float *src = dst;
    if (src_slot >= 0) {
        src = h->bases + (size_t)src_slot * h->base_stride;
        start = src_len;
        if (start == nseeds) {
            HCHECK(h, hipMemcpyAsync(dst, src, (size_t)P * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
            return 0;
        }
    } else if (powers) {   // the gpu tree's genome: root = noise[idx0] * scale_by (base.py:128-129)
        if (!h->init_scale) return h->fail("genomes with per-seed powers need dne_ga_set_init_scale first");
        const int32_t sl = slot; const int64_t o0 = seeds[0];
        int32_t *d_slot = (int32_t *)h->scratch_f;
        HCHECK(h, hipMemcpyAsync(d_slot, &sl, sizeof(sl), hipMemcpyHostToDevice, h->stream));
        HCHECK(h, hipMemcpyAsync(h->scratch_i, &o0, sizeof(o0), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_copy_noise_scaled_batch, dim3(nb, 1), dim3(256), 0, h->stream, (const float *)h->noise, (const int64_t *)h->scratch_i,
                           (const int32_t *)d_slot, h->base_stride, P, (const float *)h->init_scale, h->bases);
        HCHECK(h, hipStreamSynchronize(h->stream));   // the two scratch words are reused by the next chain
    } else {
        hipLaunchKernelGGL(k_copy_noise, dim3(nb), dim3(256), 0, h->stream, (const float *)h->noise, seeds[0], P, dst);   // ga.py:256
        launch_normc(h, dst);                                                                                                // ga.py:258-260
    }
    if (start < nseeds) {   // ga.py:262-263: all remaining mutations in one streaming pass
        const int n = nseeds - start;
        if ((size_t)n > h->chain_cap) {
            HCHECK(h, hipStreamSynchronize(h->stream));
            HCHECK(h, h->release(h->chain_offs)); HCHECK(h, h->release(h->chain_pw));
            h->chain_cap = std::max<size_t>(2 * (size_t)n, 1024);
            HCHECK(h, h->alloc(&h->chain_offs, h->chain_cap, "chain_offs")); HCHECK(h, h->alloc(&h->chain_pw, h->chain_cap, "chain_pw"));
        }
        // pageable source: the copy is staged before hipMemcpyAsync returns, so the caller's array may go away; successive
        // chains are ordered on the stream
        HCHECK(h, hipMemcpyAsync(h->chain_offs, seeds + start, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
        if (powers) HCHECK(h, hipMemcpyAsync(h->chain_pw, powers + start, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_chain_sum, dim3(nb), dim3(256), 0, h->stream, (const float *)h->noise, (const int64_t *)h->chain_offs,
                           powers ? (const float *)h->chain_pw : (const float *)nullptr, n, P, sigma, src, dst);
    }
    return 0;
}

// gpu_implementation/neuroevolution/models/base.py:190-201: the per-parameter initial scale (weights std / sqrt(fan-in) in
// dqn.py:24-27, biases 0); with it set, genomes given with per-seed powers start as noise[idx0] * scale_by
extern "C" int dne_ga_set_init_scale(dne_handle *h, const float *scale_by, size_t n) {
    DeviceGuard dg(h);
    if (h->L.kind == DNE_KIND_ES) return h->fail("dne_ga_set_init_scale needs a GA engine");
    if (n != (size_t)h->L.P) return h->fail("dne_ga_set_init_scale: expected %d values, got %zu", h->L.P, n);
    if (!h->init_scale) HCHECK(h, h->alloc(&h->init_scale, n, "init_scale"));
    HCHECK(h, hipMemcpy(h->init_scale, scale_by, n * sizeof(float), hipMemcpyHostToDevice));
    h->ga_cache.clear();
    h->child_slots.clear();          // every slot but 0 is free again -- including the ones set aside for materialised children
    h->free_slots.clear();
    for (int s2 = h->base_cap - 1; s2 >= 1; s2--) h->free_slots.push_back(s2);
    return 0;
}

extern "C" int dne_ga_rebuild(dne_handle *h, int slot, const int64_t *seeds, int nseeds, float sigma, float *out_host) {
    DeviceGuard dg(h);
    if (h->L.kind != DNE_KIND_GA) return h->fail("dne_ga_rebuild needs a GAAtariPolicy engine (LargeModel genomes carry per-seed powers: dne_ga_rebuild_powers)");
    if (slot < 0 || nseeds < 1) return h->fail("bad arguments");
    if (grow_bases(h, slot + 1)) return -1;
    h->free_slots.erase(std::remove(h->free_slots.begin(), h->free_slots.end(), slot), h->free_slots.end());
    if (build_chain(h, slot, seeds, nullptr, nseeds, sigma, -1, 0)) return -1;
    HCHECK(h, hipGetLastError());
    HCHECK(h, hipStreamSynchronize(h->stream));
    if (out_host) HCHECK(h, hipMemcpy(out_host, h->bases + (size_t)slot * h->base_stride, (size_t)h->L.P * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// the same for a genome ((idx0,), (idx1, power1), ...) of the gpu tree (base.py:118-139: compute_weights_from_seeds)
extern "C" int dne_ga_rebuild_powers(dne_handle *h, int slot, const int64_t *seeds, const float *powers, int nseeds, float *out_host) {
    DeviceGuard dg(h);
    if (h->L.kind == DNE_KIND_ES) return h->fail("dne_ga_rebuild_powers needs a GA engine");
    if (slot < 0 || nseeds < 1 || !powers) return h->fail("bad arguments");
    if (grow_bases(h, slot + 1)) return -1;
    h->free_slots.erase(std::remove(h->free_slots.begin(), h->free_slots.end(), slot), h->free_slots.end());
    if (build_chain(h, slot, seeds, powers, nseeds, 0.0f, -1, 0)) return -1;
    HCHECK(h, hipGetLastError());
    HCHECK(h, hipStreamSynchronize(h->stream));
    if (out_host) HCHECK(h, hipMemcpy(out_host, h->bases + (size_t)slot * h->base_stride, (size_t)h->L.P * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// Every child = parent chain + one fresh seed (ga.py:251-254).  Parents are materialised once into base slots (cached
// across generations by chain) and the child's last mutation is applied on the fly by the forward kernels, exactly like an
// ES perturbation with scale +sigma.  powers == null: es_distributed genomes (normc root, one sigma); powers != null: the
// gpu tree's genomes ((idx0,), (idx1, power1), ...) with a scaled-noise root (base.py:118-149).
static int ga_eval_impl(dne_handle *h, const int32_t *co, const int64_t *seeds, const float *powers, int n, float sigma, int tslimit,
                        const uint32_t *env_seed, float *returns, float *signreturns, int32_t *lengths, uint8_t *bc) {
    if (h->L.kind == DNE_KIND_ES) return h->fail("dne_ga_eval needs a GA engine");
    if (h->large && !powers) return h->fail("LargeModel genomes are the GPU tree's: per-seed powers over a scaled-noise root (dne_ga_set_init_scale + dne_ga_eval_powers)");
    if (check_n(h, n)) return -1;
    if (powers && !h->init_scale) return h->fail("genomes with per-seed powers need dne_ga_set_init_scale first");
    const int stride = powers ? 2 : 1;   // cache key: the seeds, interleaved with the bit patterns of their powers
    const int mode = powers ? 2 : 1;
    if (h->ga_cache_mode != mode || (!powers && h->ga_cache_sigma != sigma)) {   // cached parents were built under other rules
        for (auto &kv : h->ga_cache) h->free_slots.push_back(kv.second);
        h->ga_cache.clear();
        h->ga_cache_mode = mode; h->ga_cache_sigma = sigma;
    }
    auto key_of = [&](const int64_t *c, const float *pw, int len) {
        std::vector<int64_t> k((size_t)len * stride);
        for (int j = 0; j < len; j++) {
            k[(size_t)j * stride] = c[j];
            if (powers) { uint32_t bits = 0; if (j > 0) memcpy(&bits, &pw[j], 4); k[(size_t)j * stride + 1] = (int64_t)bits; }
        }
        return k;
    };
    std::vector<std::vector<int64_t>> prefix(n);
    std::vector<int64_t> off(n);
    std::vector<float> sc(n);
    std::map<std::vector<int64_t>, int> needed;
    for (int i = 0; i < n; i++) {
        const int len = co[i + 1] - co[i];
        if (len < 1) return h->fail("member %d has an empty seed chain", i);
        const int64_t *c = seeds + co[i];
        for (int j = 0; j < len; j++)
            if (check_noise_range(h, c[j], h->L.P)) return -1;
        const float *pw = powers ? powers + co[i] : nullptr;
        if (len == 1) { prefix[i] = key_of(c, pw, 1); off[i] = c[0]; sc[i] = 0.0f; }
        else { prefix[i] = key_of(c, pw, len - 1); off[i] = c[len - 1]; sc[i] = powers ? pw[len - 1] : sigma; }
        needed[prefix[i]] = -1;
    }
    size_t fresh = 0;
    for (auto &kv : needed) fresh += h->ga_cache.count(kv.first) ? 0 : 1;
    if (fresh > h->free_slots.size()) {
        if (grow_bases(h, h->base_cap + (int)(fresh - h->free_slots.size()))) return -1;
    }
    struct Fresh { const std::vector<int64_t> *chain; int slot, src_slot, src_len; };
    std::vector<Fresh> fresh_list;
    for (auto &kv : needed) {
        auto it = h->ga_cache.find(kv.first);
        if (it != h->ga_cache.end()) { kv.second = it->second; continue; }
        const int slot = h->free_slots.back();
        h->free_slots.pop_back();
        // longest cached proper prefix as the starting point
        int src_slot = -1, src_len = 0;
        std::vector<int64_t> p(kv.first);
        while ((int)p.size() > stride) {
            p.resize(p.size() - stride);
            auto jt = h->ga_cache.find(p);
            if (jt != h->ga_cache.end()) { src_slot = jt->second; src_len = (int)p.size() / stride; break; }
        }
        kv.second = slot;
        fresh_list.push_back({&kv.first, slot, src_slot, src_len});
    }
    {   // genomes with no cached ancestor: one batched launch set for all roots.  es_distributed: normc(noise[s0])
        // (ga.py:256-260); gpu tree: noise[s0] * scale_by (base.py:128-129)
        std::vector<int32_t> rslot; std::vector<int64_t> roff;
        for (auto &f : fresh_list)
            if (f.src_slot < 0) {
                if (check_noise_range(h, (*f.chain)[0], h->L.P)) return -1;
                rslot.push_back(f.slot); roff.push_back((*f.chain)[0]);
            }
        const int nr = (int)rslot.size();
        if (nr > 0) {
            if (scratch_reserve(h, (size_t)nr)) return -1;
            int32_t *d_slot = (int32_t *)h->scratch_f;
            HCHECK(h, hipMemcpyAsync(d_slot, rslot.data(), nr * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
            HCHECK(h, hipMemcpyAsync(h->scratch_i, roff.data(), nr * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
            const Layout &L = h->L;
            const size_t st = h->base_stride;
            if (powers) {
                hipLaunchKernelGGL(k_copy_noise_scaled_batch, dim3((L.P + 255) / 256, nr), dim3(256), 0, h->stream, (const float *)h->noise,
                                   (const int64_t *)h->scratch_i, (const int32_t *)d_slot, st, L.P, (const float *)h->init_scale, h->bases);
            } else {
                hipLaunchKernelGGL(k_copy_noise_batch, dim3((L.P + 255) / 256, nr), dim3(256), 0, h->stream, (const float *)h->noise,
                                   (const int64_t *)h->scratch_i, (const int32_t *)d_slot, st, L.P, h->bases);
                auto nc = [&](int off, int K, int C, float sd) { hipLaunchKernelGGL(k_normc_batch, dim3((C + 63) / 64, nr), dim3(64), 0, h->stream, h->bases, (const int32_t *)d_slot, st, off, K, C, sd); };
                auto z = [&](int off, int n2) { hipLaunchKernelGGL(k_zero_batch, dim3((n2 + 63) / 64, nr), dim3(64), 0, h->stream, h->bases, (const int32_t *)d_slot, st, off, n2); };
                nc(L.c1w, 256, 16, 1.0f); z(L.c1b, 16);
                nc(L.c2w, 256, 32, 1.0f); z(L.c2b, 32);
                nc(L.fcw, 3872, 256, 1.0f); z(L.fcb, 256);
                nc(L.ow, 256, L.nact, 0.1f); z(L.ob, L.nact);
            }
            HCHECK(h, hipStreamSynchronize(h->stream));   // scratch is reused below
        }
    }
    std::vector<int64_t> cs; std::vector<float> cp;
    for (auto &f : fresh_list) {
        const auto &key = *f.chain;
        const int len = (int)key.size() / stride;
        cs.resize(len); cp.resize(len);
        for (int j = 0; j < len; j++) {
            cs[j] = key[(size_t)j * stride];
            if (powers) { const uint32_t bits = (uint32_t)key[(size_t)j * stride + 1]; memcpy(&cp[j], &bits, 4); }
        }
        const float *pw = powers ? cp.data() : nullptr;
        if (f.src_slot < 0) {   // root already initialised; apply the remaining mutations in place
            if (len > 1 && build_chain(h, f.slot, cs.data(), pw, len, sigma, f.slot, 1)) return -1;
        } else if (build_chain(h, f.slot, cs.data(), pw, len, sigma, f.src_slot, f.src_len)) return -1;
        HCHECK(h, hipStreamSynchronize(h->stream));   // cs / cp are reused by the next chain
    }
    HCHECK(h, hipGetLastError());
    HCHECK(h, hipStreamSynchronize(h->stream));
    // keep only this generation's parents cached
    for (auto it = h->ga_cache.begin(); it != h->ga_cache.end();) {
        if (!needed.count(it->first)) { h->free_slots.push_back(it->second); it = h->ga_cache.erase(it); }
        else ++it;
    }
    for (auto &kv : needed) h->ga_cache[kv.first] = kv.second;
    std::vector<int32_t> slot(n);
    for (int i = 0; i < n; i++) slot[i] = needed[prefix[i]];
    // Children of one parent sit next to each other in the member order (DNE_GA_SORT=0 keeps the caller's order): the workgroups
    // that run side by side then stream the same parent rows, which the L2 serves once.  Members are independent, so the results
    // are the same in any order; they are handed back in the caller's.
    std::vector<int> order(n);
    for (int i = 0; i < n; i++) order[i] = i;
    if (h->ga_sort) std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return slot[a] < slot[b]; });
    std::vector<int32_t> pslot(n); std::vector<int64_t> poff(n); std::vector<float> psc(n); std::vector<uint32_t> pseed(n);
    for (int j = 0; j < n; j++) { const int i = order[j]; pslot[j] = slot[i]; poff[j] = off[i]; psc[j] = sc[i]; pseed[j] = env_seed[i]; }
    h->members_materialized = false;
    if (h->ga_materialize) {
        // Every child's vector written out once (parent + power * noise, the forward kernels' own two roundings): a member-step then
        // reads its rows as they are instead of streaming a parent row and a noise row -- for the LargeModel, whose 20 parents
        // (318 MB) do not fit the Infinity Cache, half the HBM bytes of the fc.  Roots (no mutation) are their own vector already.
        if (h->child_slots.size() < (size_t)n) {
            const int need = n - (int)h->child_slots.size();
            if ((int)h->free_slots.size() < need && grow_bases(h, h->base_cap + need - (int)h->free_slots.size())) return -1;
            for (int k = 0; k < need; k++) { h->child_slots.push_back(h->free_slots.back()); h->free_slots.pop_back(); }
        }
        std::vector<int32_t> cs(n), mp; std::vector<int64_t> mo; std::vector<float> ms; std::vector<int32_t> mc;
        for (int j = 0; j < n; j++) {
            if (psc[j] == 0.0f) { cs[j] = pslot[j]; continue; }
            cs[j] = h->child_slots[j];
            mp.push_back(pslot[j]); mo.push_back(poff[j]); ms.push_back(psc[j]); mc.push_back(cs[j]);
        }
        const int nm = (int)mp.size();
        if (nm > 0) {
            if (scratch_reserve(h, (size_t)3 * nm)) return -1;
            int32_t *d_ps = (int32_t *)h->scratch_f, *d_cs = d_ps + nm;
            float *d_sc = h->scratch_f + 2 * nm;
            HCHECK(h, hipMemcpyAsync(d_ps, mp.data(), nm * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
            HCHECK(h, hipMemcpyAsync(d_cs, mc.data(), nm * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
            HCHECK(h, hipMemcpyAsync(d_sc, ms.data(), nm * sizeof(float), hipMemcpyHostToDevice, h->stream));
            HCHECK(h, hipMemcpyAsync(h->scratch_i, mo.data(), nm * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(k_materialize_children, dim3((h->L.P + 255) / 256, nm), dim3(256), 0, h->stream, (const float *)h->noise, h->bases,
                               h->base_stride, h->L.P, (const int32_t *)d_ps, (const int64_t *)h->scratch_i, (const float *)d_sc, (const int32_t *)d_cs);
            HCHECK(h, hipStreamSynchronize(h->stream));
        }
        for (int j = 0; j < n; j++) { pslot[j] = cs[j]; psc[j] = 0.0f; }
    }
    if (dne_set_members(h, n, pslot.data(), poff.data(), psc.data())) return -1;
    h->members_materialized = h->ga_materialize != 0;   // (dne_set_members clears it: a caller's own members carry noise)
    std::vector<float> pret(n), psg(n); std::vector<int32_t> plen(n); std::vector<uint8_t> pbc(bc ? (size_t)n * 128 : 0);
    if (eval_core(h, n, 1, tslimit, pseed.data(), pret.data(), psg.data(), plen.data(), bc ? pbc.data() : nullptr)) return -1;
    for (int j = 0; j < n; j++) {
        const int i = order[j];
        returns[i] = pret[j]; lengths[i] = plen[j];
        if (signreturns) signreturns[i] = psg[j];
        if (bc) memcpy(bc + (size_t)i * 128, pbc.data() + (size_t)j * 128, 128);
    }
    return 0;
}

extern "C" int dne_ga_eval(dne_handle *h, const int32_t *co, const int64_t *seeds, int n, float sigma, int tslimit,
                           const uint32_t *env_seed, float *returns, float *signreturns, int32_t *lengths, uint8_t *bc) {
    DeviceGuard dg(h);
    return ga_eval_impl(h, co, seeds, nullptr, n, sigma, tslimit, env_seed, returns, signreturns, lengths, bc);
}

// gpu_implementation/ga.py:161-166: offspring ((idx0,), (idx1, power1), ...) -- powers[] runs parallel to seeds[] (the root's
// entry is ignored); needs dne_ga_set_init_scale
extern "C" int dne_ga_eval_powers(dne_handle *h, const int32_t *co, const int64_t *seeds, const float *powers, int n, int tslimit,
                                  const uint32_t *env_seed, float *returns, float *signreturns, int32_t *lengths, uint8_t *bc) {
    DeviceGuard dg(h);
    if (!powers) return h->fail("dne_ga_eval_powers: powers missing");
    return ga_eval_impl(h, co, seeds, powers, n, 0.0f, tslimit, env_seed, returns, signreturns, lengths, bc);
}

// ------------------------------------------------------------------------------- reduce
extern "C" int dne_centered_ranks(dne_handle *h, const float *x, int n, float *out) {
    DeviceGuard dg(h);
    if (n < 2) return h->fail("dne_centered_ranks: n = %d unsupported", n);
    if (rec_reserve(h, (n + 1) / 2, 1)) return -1;
    HCHECK(h, hipMemcpyAsync(h->scratch_f, x, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_centered_ranks, dim3((n + 255) / 256), dim3(256), 0, h->stream, (const float *)h->scratch_f, n, h->scratch_f + n);
    HCHECK(h, hipMemcpyAsync(out, h->scratch_f + n, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HCHECK(h, hipStreamSynchronize(h->stream));
    return 0;
}

static int weighted_sum_dev(dne_handle *h, const int64_t *idx_host, const float *w_dev, int n, float denom) {
    for (int i = 0; i < n; i++)
        if (check_noise_range(h, idx_host[i], h->L.P)) return -1;
    HCHECK(h, hipMemcpyAsync(h->scratch_i, idx_host, n * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HCHECK(h, hipEventRecord(h->ev_a, h->stream));
    hipLaunchKernelGGL(k_weighted_sum, dim3((h->L.P + 255) / 256), dim3(256), 0, h->stream, (const float *)h->noise,
                       (const int64_t *)h->scratch_i, w_dev, n, h->L.P, denom, h->g);
    HCHECK(h, hipEventRecord(h->ev_b, h->stream));
    HCHECK(h, hipGetLastError());
    HCHECK(h, hipStreamSynchronize(h->stream));
    float ms = 0;
    HCHECK(h, hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
    h->prof.reduce_ms = ms;
    return 0;
}

extern "C" int dne_weighted_sum(dne_handle *h, const int64_t *idx, const float *w, int n, float denom, float *g_host) {
    DeviceGuard dg(h);
    if (n < 1) return h->fail("dne_weighted_sum: n = %d unsupported", n);
    if (rec_reserve(h, n, 1)) return -1;
    HCHECK(h, hipMemcpyAsync(h->scratch_f, w, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (weighted_sum_dev(h, idx, h->scratch_f, n, denom)) return -1;
    if (g_host) HCHECK(h, hipMemcpy(g_host, h->g, (size_t)h->L.P * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int dne_optimizer_reset(dne_handle *h) {
    DeviceGuard dg(h);
    HCHECK(h, hipMemsetAsync(h->opt_m, 0, h->L.P * sizeof(float), h->stream));
    HCHECK(h, hipMemsetAsync(h->opt_v, 0, h->L.P * sizeof(float), h->stream));
    HCHECK(h, hipStreamSynchronize(h->stream));
    h->opt_t = 0;
    return 0;
}

extern "C" int dne_optimizer_get_state(dne_handle *h, float *m, float *v, int32_t *t) {
    DeviceGuard dg(h);
    HCHECK(h, hipStreamSynchronize(h->stream));
    if (m) HCHECK(h, hipMemcpy(m, h->opt_m, (size_t)h->L.P * sizeof(float), hipMemcpyDeviceToHost));
    if (v) HCHECK(h, hipMemcpy(v, h->opt_v, (size_t)h->L.P * sizeof(float), hipMemcpyDeviceToHost));
    if (t) *t = h->opt_t;
    return 0;
}

extern "C" int dne_optimizer_set_state(dne_handle *h, const float *m, const float *v, int32_t t) {
    DeviceGuard dg(h);
    if (t < 0) return h->fail("optimizer step count must be >= 0");
    if (m) HCHECK(h, hipMemcpy(h->opt_m, m, (size_t)h->L.P * sizeof(float), hipMemcpyHostToDevice));
    if (v) HCHECK(h, hipMemcpy(h->opt_v, v, (size_t)h->L.P * sizeof(float), hipMemcpyHostToDevice));
    h->opt_t = t;
    return 0;
}

extern "C" int dne_optimizer_step(dne_handle *h, int kind, float l2, double stepsize, double b1m, double b2, double eps,
                                  double *ratio) {
    DeviceGuard dg(h);
    const int P = h->L.P, nb = (P + 255) / 256;
    h->opt_t += 1;   // optimizers.py:11
    if (kind == DNE_OPT_ADAM) {
        const double a = stepsize * std::sqrt(1.0 - std::pow(b2, (double)h->opt_t)) / (1.0 - std::pow(b1m, (double)h->opt_t));
        hipLaunchKernelGGL(k_adam, dim3(nb), dim3(256), 0, h->stream, h->bases, h->opt_m, h->opt_v, (const float *)h->g, P, l2,
                           (float)(-a), (float)b1m, (float)(1.0 - b1m), (float)b2, (float)(1.0 - b2), (float)eps, h->partial);
    } else if (kind == DNE_OPT_SGD) {
        hipLaunchKernelGGL(k_sgd, dim3(nb), dim3(256), 0, h->stream, h->bases, h->opt_v, (const float *)h->g, P, l2, (float)b1m,
                           (float)(1.0 - b1m), (float)(-stepsize), h->partial);
    } else {
        return h->fail("unknown optimizer kind %d", kind);
    }
    HCHECK(h, hipGetLastError());
    std::vector<double> part(2 * (size_t)nb);
    HCHECK(h, hipMemcpyAsync(part.data(), h->partial, part.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HCHECK(h, hipStreamSynchronize(h->stream));
    double ss = 0, tt = 0;
    for (int b = 0; b < nb; b++) { ss += part[2 * b]; tt += part[2 * b + 1]; }
    if (ratio) *ratio = std::sqrt(ss) / std::sqrt(tt);
    return 0;
}

// ------------------------------------------------------------------------------- exchange
// One generation's per-pair records, gathered from every GPU, live on the device in global pair order
// (rec_idx / rec_ret / rec_sign / rec_len) and feed the update directly.
static int rec_reserve(dne_handle *h, int n_global, int per_world) {
    if ((size_t)n_global > h->rec_cap) {
        HCHECK(h, h->release(h->rec_idx)); HCHECK(h, h->release(h->rec_ret)); HCHECK(h, h->release(h->rec_sign));
        HCHECK(h, h->release(h->rec_len));
        const size_t cap = std::max<size_t>((size_t)n_global, 4096);
        HCHECK(h, h->alloc(&h->rec_idx, cap, "rec_idx")); HCHECK(h, h->alloc(&h->rec_ret, 2 * cap, "rec_ret"));
        HCHECK(h, h->alloc(&h->rec_sign, 2 * cap, "rec_sign")); HCHECK(h, h->alloc(&h->rec_len, 2 * cap, "rec_len"));
        h->rec_cap = cap;
    }
    // wire buffers: send = one shard, recv = all shards (padded to equal size), + the ordered copy for the host
    const size_t wire = (size_t)std::max(per_world, n_global) * sizeof(PairRecord);
    if (wire > h->rec_wire_cap) {
        HCHECK(h, h->release(h->rec_send)); HCHECK(h, h->release(h->rec_recv));
        HCHECK(h, h->alloc(&h->rec_send, wire, "rec_send")); HCHECK(h, h->alloc(&h->rec_recv, 2 * wire, "rec_recv"));
        h->rec_wire_cap = wire;
    }
    // scratch for processed returns [2N] + raw copy [2N] + weights [N]
    return scratch_reserve(h, (size_t)5 * n_global + 64);
}

struct Rccl {   // the few entry points of librccl.so the exchange needs
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*CommAbort)(ncclComm_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int *);
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
    void *lib = nullptr;
};
static Rccl g_rccl;

static const char *rccl_open() {   // nullptr on success, else the reason
    if (g_rccl.lib) return nullptr;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return dlerror();
#define SYM(field, name) do { *(void **)&g_rccl.field = dlsym(lib, name); if (!g_rccl.field) return "librccl.so.1 lacks " name; } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(CommAbort, "ncclCommAbort"); SYM(CommCount, "ncclCommCount"); SYM(CommUserRank, "ncclCommUserRank");
    SYM(AllGather, "ncclAllGather"); SYM(AllReduce, "ncclAllReduce"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.lib = lib;
    return nullptr;
}
#define NCHECK(h, expr)                                                                                   \
    do {                                                                                                  \
        ncclResult_t _r = (expr);                                                                         \
        if (_r != ncclSuccess) return (h)->fail("%s:%d %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r)); \
    } while (0)

extern "C" int dne_comm_unique_id(void *out128) {
    if (const char *why = rccl_open()) { g_create_error = std::string("dne_comm_unique_id: cannot open librccl.so.1: ") + why; return -1; }
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return -1; }
    memcpy(out128, &id, sizeof(id));
    return 0;
}

extern "C" int dne_comm_init(dne_handle *h, int rank, int nranks, const void *unique_id128) {
    DeviceGuard dg(h);
    if (nranks < 1 || rank < 0 || rank >= nranks) return h->fail("dne_comm_init: rank %d of %d", rank, nranks);
    if (h->comm) return h->fail("dne_comm_init: communicator already initialised");
    if (const char *why = rccl_open()) return h->fail("dne_comm_init: cannot open librccl.so.1: %s", why);
    ncclUniqueId id;
    memcpy(&id, unique_id128, sizeof(id));
    ncclComm_t c = nullptr;
    h->trace("comm init: rank %d of %d", rank, nranks);
    // the one call that can block for long (a peer that never arrives).  If the caller has meanwhile given up (dne_comm_abort from
    // the thread that owns the handle), that thread is using the handle: from here on this late path touches NOTHING of it but the
    // atomic flag -- no error text (h->fail writes the shared buffer), no trace -- and returns a code.  The handle must not be
    // destroyed while an initialisation is still in flight (bench.py leaves through os._exit in that case).
    const ncclResult_t init_rc = g_rccl.CommInitRank(&c, nranks, id, rank);
    {   // the flag is read and the communicator published under the lock dne_comm_abort takes: an abort either comes first (this
        // call drops its communicator) or finds the published one and aborts it -- never a live communicator on a handle that is off
        std::lock_guard<std::mutex> lk(h->comm_mu);
        if (h->comm_off.load(std::memory_order_acquire)) {
            if (init_rc == ncclSuccess && c) g_rccl.CommAbort(c);
            return -2;   // DNE_COMM_DROPPED
        }
        if (init_rc == ncclSuccess) { h->comm = c; h->comm_rank = rank; h->comm_size = nranks; }
    }
    if (init_rc != ncclSuccess) return h->fail("ncclCommInitRank -> %s", g_rccl.GetErrorString(init_rc));
    HCHECK(h, h->alloc(&h->comm_scratch, 64, "comm_scratch"));
    h->trace("comm ready");
    return 0;
}

extern "C" int dne_comm_info(dne_handle *h, int *rank, int *nranks, int *is_rccl) {
    int r = 0, n = 1;
    if (h->comm) {
        NCHECK(h, g_rccl.CommUserRank((ncclComm_t)h->comm, &r));
        NCHECK(h, g_rccl.CommCount((ncclComm_t)h->comm, &n));
    }
    if (rank) *rank = r;
    if (nranks) *nranks = n;
    if (is_rccl) *is_rccl = h->comm ? 1 : 0;
    return 0;
}

extern "C" int dne_device_count(int *count) {
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess && e != hipErrorNoDevice) { g_create_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); return -1; }
    *count = e == hipSuccess ? n : 0;
    return 0;
}

static void comm_destroy(dne_handle *h) {
    if (h->comm && !h->comm_borrowed) g_rccl.CommDestroy((ncclComm_t)h->comm);
    h->comm = nullptr;
}

// Several engines of one process (one per workload, all on the same device) take part in the collectives of one communicator:
// `h` borrows the communicator `owner` built with dne_comm_init.  The calls of a process are issued by one host thread, engine
// after engine, so their order is the same on every rank; the owner must outlive the borrower's last collective.
extern "C" int dne_comm_share(dne_handle *h, dne_handle *owner) {
    if (!owner || !owner->comm) return h->fail("dne_comm_share: the other handle has no communicator");
    if (h->comm) return h->fail("dne_comm_share: communicator already initialised");
    if (h->cfg.device_id != owner->cfg.device_id) return h->fail("dne_comm_share: the two handles are on different devices");
    DeviceGuard dg(h);
    h->comm = owner->comm; h->comm_rank = owner->comm_rank; h->comm_size = owner->comm_size; h->comm_borrowed = true;
    HCHECK(h, h->alloc(&h->comm_scratch, 64, "comm_scratch"));
    return 0;
}

// Give up on RCCL for this handle (the ranks agreed on another carrier): an existing communicator is aborted, one that a
// still-running dne_comm_init on another thread produces later is dropped.  Afterwards the handle behaves like a single rank.
extern "C" int dne_comm_abort(dne_handle *h) {
    std::lock_guard<std::mutex> lk(h->comm_mu);
    h->comm_off.store(true, std::memory_order_release);
    if (h->comm && !h->comm_borrowed && g_rccl.lib) g_rccl.CommAbort((ncclComm_t)h->comm);
    h->comm = nullptr; h->comm_rank = 0; h->comm_size = 1;
    return 0;
}

// element-wise sum (op 0) or max (op 1) of n <= 64 doubles over all ranks; with n = 0 it is the barrier of bench.py
extern "C" int dne_comm_allreduce(dne_handle *h, double *inout, int n, int op) {
    DeviceGuard dg(h);
    if (n < 0 || n > 64 || (op != 0 && op != 1)) return h->fail("dne_comm_allreduce: bad arguments");
    if (!h->comm) {   // a single rank: nothing to combine
        HCHECK(h, hipDeviceSynchronize());
        return 0;
    }
    const int cnt = std::max(n, 1);
    double zero = 0.0;
    HCHECK(h, hipMemcpyAsync(h->comm_scratch, n ? inout : &zero, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
    NCHECK(h, g_rccl.AllReduce(h->comm_scratch, h->comm_scratch, cnt, ncclDouble, op == 0 ? ncclSum : ncclMax, (ncclComm_t)h->comm, h->stream));
    if (n) HCHECK(h, hipMemcpyAsync(inout, h->comm_scratch, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HCHECK(h, hipStreamSynchronize(h->stream));
    HCHECK(h, hipDeviceSynchronize());
    return 0;
}

// generic small all-gather (the GA's child records, ga.py:266-271 Results): every rank contributes `bytes` bytes from the
// host and receives nranks * bytes in rank order; staged through the wire buffers, RCCL on the engine's stream
extern "C" int dne_comm_allgather(dne_handle *h, const void *send, size_t bytes, void *recv) {
    DeviceGuard dg(h);
    if (!bytes) return 0;
    if (!h->comm) { memcpy(recv, send, bytes); return 0; }
    const size_t world = (size_t)h->comm_size;
    const int as_pairs = (int)((bytes * world + sizeof(PairRecord) - 1) / sizeof(PairRecord));
    if (rec_reserve(h, 1, as_pairs)) return -1;   // wire buffers: send >= bytes, recv >= world * bytes
    HCHECK(h, hipMemcpyAsync(h->rec_send, send, bytes, hipMemcpyHostToDevice, h->stream));
    NCHECK(h, g_rccl.AllGather(h->rec_send, h->rec_recv, bytes, ncclChar, (ncclComm_t)h->comm, h->stream));
    HCHECK(h, hipMemcpyAsync(recv, h->rec_recv, bytes * world, hipMemcpyDeviceToHost, h->stream));
    HCHECK(h, hipStreamSynchronize(h->stream));
    return 0;
}

static int check_records_host(dne_handle *h, int n_global) {
    for (int i = 0; i < n_global; i++)
        if (check_noise_range(h, h->rec_idx_host[i], h->L.P)) return -1;
    return 0;
}

// this rank's shard of the last dne_es_eval as wire records (for transports other than RCCL: the redis Result path,
// the gloo tests)
extern "C" int dne_records_pack(dne_handle *h, int n_local, void *records_out) {
    DeviceGuard dg(h);
    if (h->L.kind != DNE_KIND_ES || n_local < 1 || 2 * n_local > h->M) return h->fail("dne_records_pack: %d pairs", n_local);
    if (rec_reserve(h, n_local, n_local)) return -1;
    hipLaunchKernelGGL(k_records_pack, dim3((n_local + 255) / 256), dim3(256), 0, h->stream, (const int64_t *)h->m_off, (const float *)h->ret,
                       (const float *)h->sign, (const int32_t *)h->len, n_local, n_local, (PairRecord *)h->rec_send);
    HCHECK(h, hipGetLastError());
    HCHECK(h, hipMemcpyAsync(records_out, h->rec_send, (size_t)n_local * sizeof(PairRecord), hipMemcpyDeviceToHost, h->stream));
    HCHECK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// gathered records from the host (global pair order) -> device-resident, ready for dne_es_update_gathered
extern "C" int dne_records_set(dne_handle *h, const void *records, int n_global) {
    DeviceGuard dg(h);
    if (n_global < 1) return h->fail("dne_records_set: %d pairs", n_global);
    if (rec_reserve(h, n_global, n_global)) return -1;
    HCHECK(h, hipMemcpyAsync(h->rec_recv, records, (size_t)n_global * sizeof(PairRecord), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_records_unpack, dim3((n_global + 255) / 256), dim3(256), 0, h->stream, (const PairRecord *)h->rec_recv, n_global, 1,
                       n_global, h->rec_idx, h->rec_ret, h->rec_sign, h->rec_len, (PairRecord *)nullptr);
    HCHECK(h, hipGetLastError());
    HCHECK(h, hipStreamSynchronize(h->stream));
    const PairRecord *r = (const PairRecord *)records;
    h->rec_idx_host.resize(n_global);
    for (int i = 0; i < n_global; i++) h->rec_idx_host[i] = r[i].noise_idx;
    h->rec_n = n_global;
    return check_records_host(h, n_global);
}

// The exchange step of a generation: every rank contributes the records of the n_local pairs it just evaluated
// (global pairs rank, rank + world, ...) straight from its device accumulators; one RCCL all-gather over xGMI; the
// result stays on the device in global pair order.  records_out (n_global x 32 bytes, may be NULL) receives a host copy
// for logging.  Without a communicator (one GPU) it is a local repack.
extern "C" int dne_allgather_results(dne_handle *h, int n_local, int n_global, void *records_out) {
    DeviceGuard dg(h);
    const int world = h->comm ? h->comm_size : 1, rank = h->comm ? h->comm_rank : 0;
    if (h->L.kind != DNE_KIND_ES) return h->fail("dne_allgather_results needs an ESAtariPolicy engine");
    const int mine = n_global > rank ? (n_global - rank + world - 1) / world : 0;
    if (n_global < 1 || n_local != mine || 2 * n_local > h->M)
        return h->fail("dne_allgather_results: rank %d of %d holds %d pairs, a population of %d pairs gives it %d", rank, world, n_local, n_global, mine);
    const int per = (n_global + world - 1) / world;
    if (rec_reserve(h, n_global, per * world)) return -1;
    hipLaunchKernelGGL(k_records_pack, dim3((per + 255) / 256), dim3(256), 0, h->stream, (const int64_t *)h->m_off, (const float *)h->ret,
                       (const float *)h->sign, (const int32_t *)h->len, n_local, per, (PairRecord *)h->rec_send);
    PairRecord *gathered = (PairRecord *)h->rec_recv, *ordered = gathered + (size_t)per * world;
    if (world > 1) NCHECK(h, g_rccl.AllGather(h->rec_send, gathered, (size_t)per * sizeof(PairRecord), ncclChar, (ncclComm_t)h->comm, h->stream));
    else HCHECK(h, hipMemcpyAsync(gathered, h->rec_send, (size_t)per * sizeof(PairRecord), hipMemcpyDeviceToDevice, h->stream));
    hipLaunchKernelGGL(k_records_unpack, dim3((n_global + 255) / 256), dim3(256), 0, h->stream, (const PairRecord *)gathered, n_global, world,
                       per, h->rec_idx, h->rec_ret, h->rec_sign, h->rec_len, ordered);
    HCHECK(h, hipGetLastError());
    std::vector<PairRecord> host(n_global);
    HCHECK(h, hipMemcpyAsync(host.data(), ordered, (size_t)n_global * sizeof(PairRecord), hipMemcpyDeviceToHost, h->stream));
    HCHECK(h, hipStreamSynchronize(h->stream));
    h->rec_idx_host.resize(n_global);
    for (int i = 0; i < n_global; i++) h->rec_idx_host[i] = host[i].noise_idx;
    h->rec_n = n_global;
    if (records_out) memcpy(records_out, host.data(), (size_t)n_global * sizeof(PairRecord));
    return check_records_host(h, n_global);
}

// Test hook for the one piece of the N > 1 exchange that a one-GPU box cannot reach through RCCL: the device-side un-sharding
// of a [world][per] all-gather result into global pair order.  gathered: what ncclAllGather would have produced.
extern "C" int dne_debug_unshard(dne_handle *h, const void *gathered, int n_global, int world, void *ordered_out) {
    DeviceGuard dg(h);
    if (n_global < 1 || world < 1) return h->fail("dne_debug_unshard: bad sizes");
    const int per = (n_global + world - 1) / world;
    if (rec_reserve(h, n_global, per * world)) return -1;
    PairRecord *g = (PairRecord *)h->rec_recv, *ordered = g + (size_t)per * world;
    HCHECK(h, hipMemcpyAsync(g, gathered, (size_t)per * world * sizeof(PairRecord), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_records_unpack, dim3((n_global + 255) / 256), dim3(256), 0, h->stream, (const PairRecord *)g, n_global, world, per,
                       h->rec_idx, h->rec_ret, h->rec_sign, h->rec_len, ordered);
    HCHECK(h, hipGetLastError());
    HCHECK(h, hipMemcpyAsync(ordered_out, ordered, (size_t)n_global * sizeof(PairRecord), hipMemcpyDeviceToHost, h->stream));
    HCHECK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// es.py:281-298 on the gathered, device-resident records: process returns, aggregate, optimizer step
extern "C" int dne_es_update_gathered(dne_handle *h, int proc_mode, int opt_kind, float l2, double stepsize, double b1m, double b2,
                                      double eps, double *ratio) {
    DeviceGuard dg(h);
    const int n = h->rec_n, n2 = 2 * n;
    if (n < 1) return h->fail("dne_es_update_gathered: no gathered records (dne_allgather_results / dne_records_set first)");
    float *proc = h->scratch_f, *w = h->scratch_f + n2;
    if (proc_mode == DNE_PROC_CENTERED_RANK) {          // es.py:281-282
        hipLaunchKernelGGL(k_centered_ranks, dim3((n2 + 255) / 256), dim3(256), 0, h->stream, (const float *)h->rec_ret, n2, proc);
    } else if (proc_mode == DNE_PROC_SIGN) {            // es.py:283-284
        HCHECK(h, hipMemcpyAsync(proc, h->rec_sign, n2 * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    } else if (proc_mode == DNE_PROC_CENTERED_SIGN_RANK) {   // es.py:285-286
        hipLaunchKernelGGL(k_centered_ranks, dim3((n2 + 255) / 256), dim3(256), 0, h->stream, (const float *)h->rec_sign, n2, proc);
    } else {
        return h->fail("unknown return_proc_mode %d", proc_mode);
    }
    hipLaunchKernelGGL(k_pair_weights, dim3((n + 255) / 256), dim3(256), 0, h->stream, (const float *)proc, n, w);
    HCHECK(h, hipEventRecord(h->ev_a, h->stream));
    hipLaunchKernelGGL(k_weighted_sum, dim3((h->L.P + 255) / 256), dim3(256), 0, h->stream, (const float *)h->noise,
                       (const int64_t *)h->rec_idx, (const float *)w, n, h->L.P, (float)n2, h->g);   // es.py:296 g /= returns_n2.size
    HCHECK(h, hipEventRecord(h->ev_b, h->stream));
    HCHECK(h, hipGetLastError());
    HCHECK(h, hipStreamSynchronize(h->stream));
    float ms = 0;
    HCHECK(h, hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
    h->prof.reduce_ms = ms;
    return dne_optimizer_step(h, opt_kind, l2, stepsize, b1m, b2, eps, ratio);
}

extern "C" int dne_es_update(dne_handle *h, const int64_t *idx, const float *returns_n2, const float *signreturns_n2, int n,
                             int proc_mode, int opt_kind, float l2, double stepsize, double b1m, double b2, double eps,
                             double *ratio) {
    DeviceGuard dg(h);
    if (n < 1) return h->fail("dne_es_update: n = %d unsupported", n);
    if (proc_mode != DNE_PROC_CENTERED_RANK && !signreturns_n2) return h->fail("this return_proc_mode needs signreturns_n2");
    std::vector<PairRecord> rec(n);
    for (int i = 0; i < n; i++) {
        rec[i].noise_idx = idx[i];
        rec[i].ret[0] = returns_n2[2 * i]; rec[i].ret[1] = returns_n2[2 * i + 1];
        rec[i].len[0] = rec[i].len[1] = 0;
        rec[i].sign[0] = signreturns_n2 ? signreturns_n2[2 * i] : 0.0f; rec[i].sign[1] = signreturns_n2 ? signreturns_n2[2 * i + 1] : 0.0f;
    }
    if (dne_records_set(h, rec.data(), n)) return -1;
    return dne_es_update_gathered(h, proc_mode, opt_kind, l2, stepsize, b1m, b2, eps, ratio);
}

extern "C" int dne_ga_select(dne_handle *h, const float *returns, int m, int t, int32_t *out_idx) {
    DeviceGuard dg(h);
    if (m < 1 || t < 1 || t > m) return h->fail("dne_ga_select: bad sizes m = %d t = %d", m, t);
    if (rec_reserve(h, m, 1)) return -1;
    HCHECK(h, hipMemcpyAsync(h->scratch_f, returns, m * sizeof(float), hipMemcpyHostToDevice, h->stream));
    int32_t *out = (int32_t *)(h->scratch_f + m);
    hipLaunchKernelGGL(k_ga_select, dim3((m + 255) / 256), dim3(256), 0, h->stream, (const float *)h->scratch_f, m, t, out);
    HCHECK(h, hipMemcpyAsync(out_idx, out, t * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HCHECK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ---- the novelty archive lives on the device: the master's archive only ever grows (nses.py:246-247 appends one BC per
// iteration), so a worker uploads each entry once instead of the whole archive on every call
static int archive_reserve(dne_handle *h, size_t rows, size_t entries, int dim) {
    if (rows * dim > h->arch_cap) {
        uint8_t *nb = nullptr;
        const size_t cap = std::max<size_t>(2 * rows * dim, 1u << 20);
        HCHECK(h, h->alloc(&nb, cap, "novelty_archive"));
        if (h->arch && h->arch_rows) HCHECK(h, hipMemcpy(nb, h->arch, h->arch_rows * h->arch_dim, hipMemcpyDeviceToDevice));
        HCHECK(h, h->release(h->arch));
        h->arch = nb; h->arch_cap = cap;
    }
    if (entries > h->arch_ent_cap) {
        const size_t cap = std::max<size_t>(2 * entries, 256);
        HCHECK(h, h->release(h->arch_row0)); HCHECK(h, h->release(h->arch_len));
        HCHECK(h, h->alloc(&h->arch_row0, cap, "novelty_row0")); HCHECK(h, h->alloc(&h->arch_len, cap, "novelty_len"));
        h->arch_ent_cap = cap;
        if (!h->arch_len_host.empty()) {
            HCHECK(h, hipMemcpy(h->arch_row0, h->arch_row0_host.data(), h->arch_row0_host.size() * sizeof(int64_t), hipMemcpyHostToDevice));
            HCHECK(h, hipMemcpy(h->arch_len, h->arch_len_host.data(), h->arch_len_host.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    }
    return 0;
}

extern "C" int dne_archive_clear(dne_handle *h) {
    DeviceGuard dg(h);
    HCHECK(h, hipStreamSynchronize(h->stream));
    h->arch_rows = 0; h->arch_dim = 0;
    h->arch_row0_host.clear(); h->arch_len_host.clear();
    return 0;
}

extern "C" int dne_archive_append(dne_handle *h, const uint8_t *bc, int bc_len, int dim) {
    DeviceGuard dg(h);
    if (bc_len < 1 || dim < 1) return h->fail("dne_archive_append: empty entry");
    if (h->arch_dim && dim != h->arch_dim) return h->fail("dne_archive_append: entry width %d, archive holds %d", dim, h->arch_dim);
    HCHECK(h, hipStreamSynchronize(h->stream));
    const size_t n = h->arch_len_host.size();
    if (archive_reserve(h, h->arch_rows + bc_len, n + 1, dim)) return -1;
    h->arch_dim = dim;
    if (copy_h2d(h, h->arch + h->arch_rows * dim, bc, (size_t)bc_len * dim)) return -1;
    const int64_t row0 = (int64_t)h->arch_rows; const int32_t len = bc_len;
    HCHECK(h, hipMemcpy(h->arch_row0 + n, &row0, sizeof(row0), hipMemcpyHostToDevice));
    HCHECK(h, hipMemcpy(h->arch_len + n, &len, sizeof(len), hipMemcpyHostToDevice));
    h->arch_row0_host.push_back(row0); h->arch_len_host.push_back(len);
    h->arch_rows += bc_len;
    return 0;
}

extern "C" int dne_archive_size(dne_handle *h) { return (int)h->arch_len_host.size(); }

// a caller-supplied archive replaces the resident one (the one-shot form of the two novelty calls)
static int archive_load(dne_handle *h, const uint8_t *archive, const int32_t *alen, int narch, int dim) {
    if (narch < 1) return h->fail("empty archive");
    if (dne_archive_clear(h)) return -1;
    size_t rows = 0;
    for (int a = 0; a < narch; a++) { if (alen[a] < 1) return h->fail("empty archive entry"); rows += alen[a]; }
    if (archive_reserve(h, rows, narch, dim)) return -1;
    h->arch_dim = dim;
    if (copy_h2d(h, h->arch, archive, rows * dim)) return -1;
    int64_t r0 = 0;
    for (int a = 0; a < narch; a++) { h->arch_row0_host.push_back(r0); h->arch_len_host.push_back(alen[a]); r0 += alen[a]; }
    HCHECK(h, hipMemcpy(h->arch_row0, h->arch_row0_host.data(), narch * sizeof(int64_t), hipMemcpyHostToDevice));
    HCHECK(h, hipMemcpy(h->arch_len, h->arch_len_host.data(), narch * sizeof(int32_t), hipMemcpyHostToDevice));
    h->arch_rows = rows;
    return 0;
}

static int novelty_scratch(dne_handle *h, size_t out_words, size_t len_words) {
    if (out_words > h->nov_out_cap) {
        HCHECK(h, h->release(h->nov_out));
        h->nov_out_cap = std::max<size_t>(2 * out_words, 4096);
        HCHECK(h, h->alloc(&h->nov_out, h->nov_out_cap, "novelty_out"));
    }
    if (len_words > h->nov_len_cap) {
        HCHECK(h, h->release(h->nov_len));
        h->nov_len_cap = std::max<size_t>(2 * len_words, 4096);
        HCHECK(h, h->alloc(&h->nov_len, h->nov_len_cap, "novelty_len_in"));
    }
    return 0;
}

// nses.py:12-32.  archive == NULL: score against the device-resident archive (dne_archive_append)
extern "C" int dne_novelty(dne_handle *h, const uint8_t *archive, const int32_t *alen, int narch, const uint8_t *bc,
                           int bc_len, int dim, int k, double *out) {
    DeviceGuard dg(h);
    if (bc_len < 1 || dim < 1 || k < 1) return h->fail("dne_novelty: bad sizes");
    if (archive && archive_load(h, archive, alen, narch, dim)) return -1;
    narch = (int)h->arch_len_host.size();
    if (narch < 1) return h->fail("dne_novelty: the archive is empty");
    if (dim != h->arch_dim) return h->fail("dne_novelty: characterisation width %d, archive holds %d", dim, h->arch_dim);
    if (novelty_scratch(h, 2 * (size_t)narch, ((size_t)bc_len * dim + 3) / 4)) return -1;
    uint8_t *d_bc = (uint8_t *)h->nov_len;
    HCHECK(h, hipMemcpyAsync(d_bc, bc, (size_t)bc_len * dim, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_bc_sqdist, dim3(narch), dim3(256), 0, h->stream, (const uint8_t *)h->arch, (const int64_t *)h->arch_row0,
                       (const int32_t *)h->arch_len, (const uint8_t *)d_bc, bc_len, dim, h->nov_out);
    HCHECK(h, hipGetLastError());
    std::vector<long long> ab(2 * (size_t)narch);
    HCHECK(h, hipMemcpyAsync(ab.data(), h->nov_out, ab.size() * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    HCHECK(h, hipStreamSynchronize(h->stream));
    std::vector<double> d(narch);
    for (int a = 0; a < narch; a++) {   // nses.py:20 sqrt(a**2 + b**2) with a, b = the two Frobenius norms
        const double na = std::sqrt((double)ab[2 * a]), nb = std::sqrt((double)ab[2 * a + 1]);
        d[a] = std::sqrt(na * na + nb * nb);
    }
    std::sort(d.begin(), d.end());        // nses.py:29-31 k nearest, mean
    const int kk = std::min(k, narch);
    double s = 0;
    for (int i = 0; i < kk; i++) s += d[i];
    *out = s / kk;
    return 0;
}

extern "C" int dne_novelty_batch(dne_handle *h, const uint8_t *archive, const int32_t *alen, int narch, int n,
                                 const int32_t *lengths, int k, double *out) {
    DeviceGuard dg(h);
    if (h->L.kind != DNE_KIND_ES || !h->bc || h->cfg.bc_final_only) return h->fail("dne_novelty_batch needs an ES engine created with record_bc = 1 (full trajectories)");
    if (check_n(h, n)) return -1;
    if (k < 1) return h->fail("dne_novelty_batch: bad sizes");
    if (archive && archive_load(h, archive, alen, narch, 128)) return -1;
    narch = (int)h->arch_len_host.size();
    if (narch < 1) return h->fail("dne_novelty_batch: the archive is empty");
    if (h->arch_dim != 128) return h->fail("dne_novelty_batch: the archive must hold 128-byte RAM rows");
    for (int i = 0; i < n; i++)
        if (lengths[i] < 1 || lengths[i] > h->cfg.bc_max_steps) return h->fail("member %d: trajectory length %d outside the recorded capacity %d", i, lengths[i], h->cfg.bc_max_steps);
    if (novelty_scratch(h, (size_t)n * narch * 2, n)) return -1;
    HCHECK(h, hipMemcpyAsync(h->nov_len, lengths, n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_bc_sqdist_batch, dim3(narch, n), dim3(256), 0, h->stream, (const uint8_t *)h->arch, (const int64_t *)h->arch_row0,
                       (const int32_t *)h->arch_len, (const uint8_t *)h->bc, (const int32_t *)h->nov_len, h->cfg.bc_max_steps, narch, h->nov_out);
    HCHECK(h, hipGetLastError());
    std::vector<long long> ab((size_t)n * narch * 2);
    HCHECK(h, hipMemcpyAsync(ab.data(), h->nov_out, ab.size() * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    HCHECK(h, hipStreamSynchronize(h->stream));
    std::vector<double> d(narch);
    const int kk = std::min(k, narch);
    for (int i = 0; i < n; i++) {
        for (int a = 0; a < narch; a++) {   // nses.py:12-20
            const double na = std::sqrt((double)ab[((size_t)i * narch + a) * 2]), nb = std::sqrt((double)ab[((size_t)i * narch + a) * 2 + 1]);
            d[a] = std::sqrt(na * na + nb * nb);
        }
        std::sort(d.begin(), d.end());      // nses.py:29-31
        double s = 0;
        for (int j = 0; j < kk; j++) s += d[j];
        out[i] = s / kk;
    }
    return 0;
}
