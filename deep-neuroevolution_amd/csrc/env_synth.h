// env_synth.h -- device-resident batched SynthAtari stepper + fused DeepMind preprocessing.
//
// Replaces, for the hot path, gym/ALE + es_distributed/atari_wrappers.py:204-222 (wrap_deepmind):
//   NoopResetEnv :18-31, MaxAndSkipEnv :95-107 (4 raw frames, reward sum, max over the last 2 frames),
//   FireResetEnv :40-48, WarpFrame :138-142 (gray -> PIL antialiased bilinear 84x84 -> u8 truncation),
//   FrameStack :167-180 (4 frames on the channel axis), ScaledFloatFrame :183-186 (/255, fused into conv1).
// The emulator itself is the SynthAtari fixture specified in DESIGN.md (ALE and ROMs are not available);
// this file is an independent implementation of that written spec (the CPU oracle is another one).
//
// One lane advances a member's emulator state in registers (struct Emu); a workgroup per member then renders
// max(frame_prev, frame_cur) as a byte image of colour pairs in LDS, runs PIL's two separable passes
// (double accumulation, float32 intermediate, horizontal first) band by band, and shifts the new u8
// frame into the member's [84][84][4] stack with one dword read-modify-write per pixel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dne {

// Profiling build only (make clock -> libdne_hip_clock.so, tools/phase_clock.py): thread 0 of the first 128 workgroups of the
// tail's kernels leaves the 100 MHz wall clock at a few milestones of its last launch.  Compiled out of the product library.
#ifdef DNE_PHASE_CLOCK
__device__ long long g_phase[6][128][8];
#define DNE_PHASE(K, I) do { if (threadIdx.x == 0 && blockIdx.x < 128) dne::g_phase[K][blockIdx.x][I] = (long long)wall_clock64(); } while (0)
// accumulated form for kernels that loop over frames / stages: time between consecutive DNE_ACC points, summed per slot
#define DNE_ACC_DECL long long pacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long pprev_ = (long long)wall_clock64(); pacc_[7] = pprev_   /* slot 7: the workgroup's start */
#define DNE_ACC(I) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = (long long)wall_clock64(); pacc_[I] += t_ - pprev_; pprev_ = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define DNE_ACC_STORE(K) DNE_ACC_STORE_EVERY(K, 1)
// every STRIDE-th workgroup of the launch (the first 128 of them): a sample over the whole launch instead of its first wave; slot 6: the end
// (a launch of fewer than 128 workgroups: stride 1, not 0)
#define DNE_ACC_STORE_EVERY(K, STRIDE) do { const unsigned st_ = (STRIDE) > 0 ? (unsigned)(STRIDE) : 1u; if (threadIdx.x == 0 && blockIdx.x % st_ == 0 && blockIdx.x / st_ < 128) { pacc_[6] = (long long)wall_clock64(); for (int i_ = 0; i_ < 8; i_++) dne::g_phase[K][blockIdx.x / st_][i_] = pacc_[i_]; } } while (0)
// k_fc_duo's table timeline (tools/duo_tick_clock.py): for 64 workgroups spread over the launch, the second work item of each --
// per wave and tick the shader clock in front of and behind the tick's s_barrier, plus the item's plan (start delay and length of
// every wave in ticks) and the 100 MHz wall clock at both ends (calibrates the shader clock)
constexpr int DUO_TICK_WGS = 64, DUO_TICK_MAX = 288;
__device__ long long g_duo_tick[DUO_TICK_WGS][8][DUO_TICK_MAX][2];   // (k_fc_duo: waves 0-3; k_fc_ring: all eight)
__device__ long long g_duo_plan[DUO_TICK_WGS][8][8];   // delay, len, tmax, ticks recorded, memtime start, memtime end, wall start, wall end
// Round 6: one record per WORKGROUP of the lock-step's kernels (tools/wg_clock.py): kernel id, block, 100 MHz wall clock at the
// workgroup's start and end, and where it ran (HW_ID: CU / SE / SIMD of wave 0, XCC_ID) -- whether a kernel that is slow beside the
// streaming fc is slow per workgroup (shared pipes) or starts its workgroups late (no room on the CUs).  A ring of WGCLK_CAP records.
constexpr unsigned WGCLK_CAP = 1u << 19;
__device__ long long g_wgclk[WGCLK_CAP][4];
__device__ unsigned g_wgclk_n;
#define DNE_WG_BEGIN const long long wg_t0_ = (long long)wall_clock64()
#define DNE_WG_END(K) do { if (threadIdx.x == 0) { unsigned hw_, xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_)); const unsigned i_ = atomicAdd(&dne::g_wgclk_n, 1u) % dne::WGCLK_CAP; \
    dne::g_wgclk[i_][0] = ((long long)(K) << 32) | (long long)blockIdx.x; dne::g_wgclk[i_][1] = wg_t0_; dne::g_wgclk[i_][2] = (long long)wall_clock64(); \
    dne::g_wgclk[i_][3] = ((long long)xcc_ << 32) | (long long)hw_; } } while (0)
#else
#define DNE_WG_BEGIN do { } while (0)
#define DNE_WG_END(K) do { } while (0)
#define DNE_PHASE(K, I) do { } while (0)
#define DNE_ACC_DECL do { } while (0)
#define DNE_ACC(I) do { } while (0)
#define DNE_ACC_STORE(K) do { } while (0)
#define DNE_ACC_STORE_EVERY(K, STRIDE) do { } while (0)
#endif

// RAM map (DESIGN.md "SynthAtari")
enum : int {
    RM_FC = 0, RM_RNG = 2, RM_PX = 6, RM_PROW = 7, RM_LIVES = 8, RM_OVER = 9, RM_TEMP = 10, RM_COOL = 11,
    RM_OFF = 12, RM_VIS = 16, RM_IGLOO = 20, RM_LEVEL = 21, RM_SCORE = 22, RM_FREEZE = 25, RM_HZX = 26,
    RM_HZA = 30, RM_DIR = 34, RM_LASTA = 38, RM_TICK = 39
};

constexpr int RAM_LIVE = 40;   // highest used RAM byte is RM_TICK = 39

// The emulator state lives in registers while a member is stepped: the per-row fields (floe offset, visited flag,
// hazard x / active, direction) packed four bytes to a dword and selected with a shift, everything else as scalars.
// emu_unpack / emu_pack convert from / to the first 10 dwords of the 128-byte RAM (DESIGN.md "SynthAtari" map).
struct Emu {
    uint32_t rng, off4, vis4, hzx4, hza4, dir4;
    int fc, px, prow, lives, over, temp, cool, igloo, level, score, freeze, lasta, tick;
};

__device__ __forceinline__ uint32_t emu_b4(uint32_t w, int r) { return (w >> (8 * r)) & 255u; }
__device__ __forceinline__ uint32_t emu_set4(uint32_t w, int r, uint32_t v) {
    const int sh = 8 * r;
    return (w & ~(255u << sh)) | ((v & 255u) << sh);
}

__device__ __forceinline__ Emu emu_unpack(const uint32_t *w) {
    Emu e;
    e.fc = w[0] & 0xffffu;
    e.rng = (w[0] >> 16) | (w[1] << 16);
    e.px = (w[1] >> 16) & 255u; e.prow = w[1] >> 24;
    e.lives = w[2] & 255u; e.over = (w[2] >> 8) & 255u; e.temp = (w[2] >> 16) & 255u; e.cool = w[2] >> 24;
    e.off4 = w[3]; e.vis4 = w[4];
    e.igloo = w[5] & 255u; e.level = (w[5] >> 8) & 255u;
    e.score = (w[5] >> 16) | ((w[6] & 255u) << 16);
    e.freeze = (w[6] >> 8) & 255u;
    e.hzx4 = (w[6] >> 16) | (w[7] << 16);
    e.hza4 = (w[7] >> 16) | (w[8] << 16);
    e.dir4 = (w[8] >> 16) | (w[9] << 16);
    e.lasta = (w[9] >> 16) & 255u; e.tick = w[9] >> 24;
    return e;
}

__device__ __forceinline__ void emu_pack(const Emu &e, uint32_t *w) {
    w[0] = (uint32_t)e.fc | (e.rng << 16);
    w[1] = (e.rng >> 16) | ((uint32_t)e.px << 16) | ((uint32_t)e.prow << 24);
    w[2] = (uint32_t)e.lives | ((uint32_t)e.over << 8) | ((uint32_t)e.temp << 16) | ((uint32_t)e.cool << 24);
    w[3] = e.off4; w[4] = e.vis4;
    w[5] = (uint32_t)e.igloo | ((uint32_t)e.level << 8) | (((uint32_t)e.score & 0xffffu) << 16);
    w[6] = ((uint32_t)e.score >> 16) | ((uint32_t)e.freeze << 8) | (e.hzx4 << 16);
    w[7] = (e.hzx4 >> 16) | (e.hza4 << 16);
    w[8] = (e.hza4 >> 16) | (e.dir4 << 16);
    w[9] = (e.dir4 >> 16) | ((uint32_t)e.lasta << 16) | ((uint32_t)e.tick << 24);
}

__device__ __forceinline__ uint32_t emu_rand(Emu &e) {
    e.rng = e.rng * 1664525u + 1013904223u;
    return e.rng >> 16;
}

__device__ __forceinline__ Emu emu_reset(uint32_t seed) {
    Emu e{};
    e.rng = seed ^ 0x9E3779B9u;
    e.px = 76; e.lives = 3; e.temp = 45;
#pragma unroll
    for (int r = 0; r < 4; r++) { e.off4 = emu_set4(e.off4, r, emu_rand(e) % 160u); e.dir4 = emu_set4(e.dir4, r, r & 1); }
#pragma unroll
    for (int r = 0; r < 4; r++) e.hzx4 = emu_set4(e.hzx4, r, emu_rand(e) % 160u);
    return e;
}

__device__ __forceinline__ bool emu_on_floe(const Emu &e, int px, int r) {
    const int rel = (px + 164 - (int)emu_b4(e.off4, r)) % 160;
    return (rel % 40) < 32;
}

// one raw emulator frame; returns the integer reward
__device__ inline int emu_frame(Emu &e, int a) {
    if (e.over) return 0;
    e.fc = (e.fc + 1) & 0xffff;
    const int fc = e.fc;
    e.lasta = a;
    int level = e.level;
    const int speed = level >= 3 ? 2 : 1;
    int reward = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int o = emu_b4(e.off4, r);
        e.off4 = emu_set4(e.off4, r, emu_b4(e.dir4, r) ? (o + 160 - speed) % 160 : (o + speed) % 160);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (emu_b4(e.hza4, r)) {
            const int x = emu_b4(e.hzx4, r);
            e.hzx4 = emu_set4(e.hzx4, r, emu_b4(e.dir4, r) ? (x + 1) % 160 : (x + 159) % 160);
        } else if ((fc & 63) == 16 * r) {
            if ((emu_rand(e) & 3u) == 0u) { e.hza4 = emu_set4(e.hza4, r, 1); e.hzx4 = emu_set4(e.hzx4, r, emu_b4(e.dir4, r) ? 0 : 159); }
        }
    }
    // ALE action set: NOOP FIRE UP RIGHT LEFT DOWN UR UL DR DL UF RF LF DF URF ULF DRF DLF
    const int dx = (0x14948 >> a) & 1 ? 1 : ((0x29290 >> a) & 1 ? -1 : 0);   // RIGHT-ish bits / LEFT-ish bits
    const int dy = (0x0C4C4 >> a) & 1 ? -1 : ((0x32320 >> a) & 1 ? 1 : 0);   // UP-ish / DOWN-ish
    const bool fire = a == 1 || a >= 10;
    bool died = false;
    if (e.freeze > 0) {
        e.freeze--;
    } else {
        int px = e.px, prow = e.prow;
        if (prow > 0) px = (px + (emu_b4(e.dir4, prow - 1) ? -speed : speed) + 2 * dx + 160) % 160;   // ice rows wrap
        else { px += 2 * dx; px = px < 8 ? 8 : (px > 144 ? 144 : px); }
        if (e.cool > 0) {
            e.cool--;
        } else if (dy != 0) {
            const int tgt = prow + dy;
            if (tgt < 0) {
                if (e.igloo >= 16 && px >= 104) {   // enter the finished igloo: level complete
                    reward += 10 * e.temp + 100;
                    if (level < 255) level++;
                    e.level = level; e.igloo = 0; e.temp = 45; e.tick = 0;
                    e.vis4 = 0; e.hza4 = 0;
                    prow = 0; px = 76; e.freeze = 64;
                }
            } else if (tgt <= 4) {
                prow = tgt;
                e.cool = 12;
                if (prow == 0) px = px < 8 ? 8 : (px > 144 ? 144 : px);
                if (prow > 0) {
                    const int r = prow - 1;
                    if (emu_on_floe(e, px, r)) {
                        if (!emu_b4(e.vis4, r)) {
                            e.vis4 = emu_set4(e.vis4, r, 1);
                            reward += 10;
                            if (e.igloo < 16) e.igloo++;
                            if (e.vis4 & (e.vis4 >> 8) & (e.vis4 >> 16) & (e.vis4 >> 24) & 255u) e.vis4 = 0;
                        }
                    } else {
                        died = true;
                    }
                }
            }
        } else if (fire && prow > 0 && e.igloo > 0) {
            e.dir4 ^= 1u << (8 * (prow - 1)); e.igloo--; e.cool = 12;
        }
        if (!died && prow > 0) {
            const int r = prow - 1;
            if (!emu_on_floe(e, px, r)) died = true;
            else if (emu_b4(e.hza4, r)) {
                const int d = px + 4 - (int)emu_b4(e.hzx4, r);
                if ((d < 0 ? -d : d) < 8) died = true;
            }
        }
        e.px = px; e.prow = prow;
    }
    if (++e.tick >= 48) {
        e.tick = 0;
        if (e.temp > 0) e.temp--;
        if (e.temp == 0) died = true;
    }
    if (died) {
        if (e.lives == 0) e.over = 1; else e.lives--;
        e.prow = 0; e.px = 76; e.freeze = 128; e.cool = 0;
        e.hza4 = 0;
        if (e.temp == 0) e.temp = 45;
    }
    if (reward) e.score = (e.score + reward / 10) & 0xffffff;
    return reward;
}

// Everything the renderer needs from one RAM snapshot, unpacked once per thread into registers
// (per-row fields stay packed four-to-a-dword and are selected with a shift).
struct PixState {
    int px, py, temp2, lives10, score, igloo, sky;
    uint32_t off4, vis4, hzx4, hza4;
    bool blink;
};

__device__ __forceinline__ PixState synth_pix_state(const uint8_t *ram) {
    PixState p;
    const int prow = ram[RM_PROW];
    p.px = ram[RM_PX];
    p.py = prow == 0 ? 62 : 48 + 32 * prow;
    p.blink = ram[RM_FREEZE] > 0 && (ram[RM_FC] & 4);
    p.temp2 = 8 + 2 * ram[RM_TEMP];
    p.lives10 = 10 * ram[RM_LIVES];
    p.score = ram[RM_SCORE] | (ram[RM_SCORE + 1] << 8);
    p.igloo = ram[RM_IGLOO];
    p.sky = (ram[RM_LEVEL] & 1) ? 3 : 2;
    p.off4 = *(const uint32_t *)(ram + RM_OFF);
    p.vis4 = *(const uint32_t *)(ram + RM_VIS);
    p.hzx4 = ram[RM_HZX] | (ram[RM_HZX + 1] << 8) | (ram[RM_HZX + 2] << 16) | ((uint32_t)ram[RM_HZX + 3] << 24);
    p.hza4 = ram[RM_HZA] | (ram[RM_HZA + 1] << 8) | (ram[RM_HZA + 2] << 16) | ((uint32_t)ram[RM_HZA + 3] << 24);
    return p;
}

// palette index of screen pixel (x, y)
__device__ __forceinline__ int synth_pixel(const PixState &p, int x, int y) {
    if (!p.blink && (unsigned)(y - p.py) < 16u) {
        int d = x - p.px;
        d += d < 0 ? 160 : 0;
        if (d < 8) return 8;
    }
    if (y < 8 || y >= 208) return 0;
    if (y < 16) {
        if (x >= 8 && x < p.temp2) return 11;
        const int q = x - 120;
        if (q >= 0 && q < p.lives10 && (q % 10) < 6) return 12;
        return 1;
    }
    if (y < 20) {
        if (x >= 8 && x < 136 && ((x - 8) & 7) < 6 && ((p.score >> ((x - 8) >> 3)) & 1)) return 14;
        return 1;
    }
    if (y < 64) {
        if (x >= 112 && x < 144 && y >= 40) {
            if (p.igloo >= 16 && x >= 124 && x < 132 && y >= 52) return 13;
            if (((63 - y) / 6) * 4 + ((x - 112) >> 3) < p.igloo) return 10;
        }
        return p.sky;
    }
    if (y < 80) return 4;
    const int r8 = ((y - 80) >> 5) * 8, yo = (y - 80) & 31;
    if (yo >= 4 && yo < 12 && ((p.hza4 >> r8) & 255u)) {
        const int d = x - (int)((p.hzx4 >> r8) & 255u);
        if ((d < 0 ? -d : d) < 6) return 9;
    }
    if (yo >= 16 && yo < 28) {
        int rel = x - (int)((p.off4 >> r8) & 255u);
        rel += rel < 0 ? 160 : 0;
        if ((rel % 40) < 32) return ((p.vis4 >> r8) & 255u) ? 7 : 6;
    }
    return (yo & 8) ? 15 : 5;
}

// the state is the same in every lane: move it to scalar registers
__device__ __forceinline__ PixState synth_pix_uniform(const PixState &q) {
    PixState p;
    p.px = __builtin_amdgcn_readfirstlane(q.px); p.py = __builtin_amdgcn_readfirstlane(q.py);
    p.temp2 = __builtin_amdgcn_readfirstlane(q.temp2); p.lives10 = __builtin_amdgcn_readfirstlane(q.lives10);
    p.score = __builtin_amdgcn_readfirstlane(q.score); p.igloo = __builtin_amdgcn_readfirstlane(q.igloo);
    p.sky = __builtin_amdgcn_readfirstlane(q.sky);
    p.off4 = __builtin_amdgcn_readfirstlane(q.off4); p.vis4 = __builtin_amdgcn_readfirstlane(q.vis4);
    p.hzx4 = __builtin_amdgcn_readfirstlane(q.hzx4); p.hza4 = __builtin_amdgcn_readfirstlane(q.hza4);
    p.blink = __builtin_amdgcn_readfirstlane((int)q.blink) != 0;
    return p;
}

// One screen row of one frame as the renderer paints it: background, one periodic pattern, one span above it, the
// player sprite on top.  Every row class of synth_pixel fits this shape.  All values are wave-uniform and are kept
// as separate scalars (a struct here ends up in scratch memory: the colour selects turn into indexed loads).
//   pattern: rel = (x - p_off) mod 160 hits when rel < p_len, rel % p_period < p_width and bit rel / p_period of p_mask
//            is set (p_recip = ceil(65536 / p_period));  span: [r_lo, r_lo + r_len);  sprite: 8 pixels from s_px, wrapping
#define DNE_ROW_FIELDS(pre)                                                                                               \
    int pre##bg, pre##p_off, pre##p_len, pre##p_period, pre##p_recip, pre##p_width, pre##p_col, pre##r_lo, pre##r_len,    \
        pre##r_col, pre##s_on, pre##s_px;                                                                                 \
    uint32_t pre##p_mask
#define DNE_ROW_ARGS(pre)                                                                                                 \
    pre##bg, pre##p_off, pre##p_len, pre##p_period, pre##p_recip, pre##p_width, pre##p_col, pre##p_mask, pre##r_lo,       \
        pre##r_len, pre##r_col, pre##s_on, pre##s_px

__device__ __forceinline__ void synth_row_desc(const PixState &p, int y, int &bg, int &p_off, int &p_len, int &p_period,
                                               int &p_recip, int &p_width, int &p_col, uint32_t &p_mask, int &r_lo, int &r_len,
                                               int &r_col, int &s_on, int &s_px) {
    // straight-line selects on the scalar unit
    const bool hud = y >= 8 && y < 16, sco = y >= 16 && y < 20, sky = y >= 20 && y < 64, shore = y >= 64 && y < 80;
    const bool wat = y >= 80 && y < 208;
    const int yw = wat ? y - 80 : 0, r8 = (yw >> 5) * 8, yo = yw & 31;
    const bool floe = wat && yo >= 16 && yo < 28;
    const bool haz = wat && yo >= 4 && yo < 12 && ((p.hza4 >> r8) & 255u) != 0u;
    const bool igl = sky && y >= 40, door = igl && y >= 52 && p.igloo >= 16;
    int nb = p.igloo - ((63 - (igl ? y : 63)) / 6) * 4;
    nb = nb < 0 ? 0 : nb > 4 ? 4 : nb;
    s_on = (!p.blink && (unsigned)(y - p.py) < 16u) ? 1 : 0;
    s_px = p.px;
    bg = (hud || sco) ? 1 : sky ? p.sky : shore ? 4 : wat ? ((yo & 8) ? 15 : 5) : 0;
    p_off = hud ? 120 : sco ? 8 : sky ? 112 : (int)((p.off4 >> r8) & 255u);
    p_len = hud ? p.lives10 : sco ? 128 : igl ? 32 : floe ? 160 : 0;
    p_period = hud ? 10 : wat ? 40 : 8;
    p_recip = hud ? 6554 : wat ? 1639 : 8192;
    p_width = (hud || sco) ? 6 : sky ? 8 : 32;
    p_col = hud ? 12 : sco ? 14 : sky ? 10 : (((p.vis4 >> r8) & 255u) ? 7 : 6);
    p_mask = sco ? (uint32_t)p.score : sky ? (1u << nb) - 1u : 0xffffffffu;
    r_lo = hud ? 8 : sky ? 124 : (int)((p.hzx4 >> r8) & 255u) - 5;
    r_len = hud ? p.temp2 - 8 : door ? 8 : haz ? 11 : 0;
    r_col = hud ? 11 : sky ? 13 : 9;
}

// the three layers of a row, each for one pixel x: the caller skips a layer the row does not have (wave-uniform tests on the description)
__device__ __forceinline__ int synth_row_pattern(int x, int c, int p_off, int p_len, int p_period, int p_recip, int p_width, int p_col, uint32_t p_mask) {
    int rel = x - p_off;
    rel += rel < 0 ? 160 : 0;
    // (24-bit multiplies: rel < 160, p_recip <= 8192, cell <= 20, p_period <= 40 -- v_mul_lo_u32 runs at a quarter of v_mul_u32_u24's rate)
    const int cell = (int)(__umul24((unsigned)rel, (unsigned)p_recip) >> 16);   // rel / p_period for 0 <= rel < 160 and the periods 8, 10, 40
    const int rem = rel - (int)__umul24((unsigned)cell, (unsigned)p_period);
    const bool hit = (rel < p_len) & (rem < p_width) & (((p_mask >> cell) & 1u) != 0u);
    return hit ? p_col : c;
}
__device__ __forceinline__ int synth_row_span(int x, int c, int r_lo, int r_len, int r_col) { return (unsigned)(x - r_lo) < (unsigned)r_len ? r_col : c; }
__device__ __forceinline__ int synth_row_sprite(int x, int c, int s_px) {
    int sx = x - s_px;
    sx += sx < 0 ? 160 : 0;
    return sx < 8 ? 8 : c;
}

__device__ __forceinline__ int synth_row_pixel(int x, int bg, int p_off, int p_len, int p_period, int p_recip, int p_width,
                                               int p_col, uint32_t p_mask, int r_lo, int r_len, int r_col, int s_on, int s_px) {
    int c = synth_row_pattern(x, bg, p_off, p_len, p_period, p_recip, p_width, p_col, p_mask);
    c = synth_row_span(x, c, r_lo, r_len, r_col);
    return s_on != 0 ? synth_row_sprite(x, c, s_px) : c;
}

// pixels lane, lane + 64, lane + 128 of a row from its description (three int4 words, the same in every lane).  Most rows have no pattern
// (shore, sky above the igloo, 20 of a water band's 32 rows), no span and no sprite: an empty pattern (p_len <= 0), an empty span (r_len <= 0)
// and a sprite that is off change nothing, so those layers are skipped as a whole -- tested once per row on the scalar unit.
__device__ __forceinline__ void synth_row_paint3(int lane, const int4 &q0, const int4 &q1, const int4 &q2, int &c0, int &c1, int &c2) {
    c0 = c1 = c2 = q0.x;
    if (__builtin_amdgcn_readfirstlane(q0.z) > 0) {
        c0 = synth_row_pattern(lane, c0, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, (uint32_t)q1.w);
        c1 = synth_row_pattern(lane + 64, c1, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, (uint32_t)q1.w);
        c2 = synth_row_pattern(lane + 128, c2, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, (uint32_t)q1.w);
    }
    if (__builtin_amdgcn_readfirstlane(q2.y) > 0) {
        c0 = synth_row_span(lane, c0, q2.x, q2.y, q2.z);
        c1 = synth_row_span(lane + 64, c1, q2.x, q2.y, q2.z);
        c2 = synth_row_span(lane + 128, c2, q2.x, q2.y, q2.z);
    }
    if (__builtin_amdgcn_readfirstlane(q2.w) & 1) {
        c0 = synth_row_sprite(lane, c0, q2.w >> 1);
        c1 = synth_row_sprite(lane + 64, c1, q2.w >> 1);
        c2 = synth_row_sprite(lane + 128, c2, q2.w >> 1);
    }
}

// Screen rows fall into 45 static classes (HUD bands, sky, igloo block rows, shore, 4-row strips of the
// water) inside which every row has identical pixels, except where the 16-row player sprite of the
// previous / current frame overlaps.  Only one representative row per (class, player-in-prev,
// player-in-cur) key is rendered and horizontally resized (~50 of 210 rows); the vertical pass reads
// through the row -> slot map.  Identical rows give bit-identical results, so this is exact.
constexpr int ENV_MAX_ROWS = 80;   // 45 classes + at most 3 extra keys for each of the <= 10 classes a sprite touches

// PIL's BILINEAR windows for 160 -> 84 hold 3 or 4 source pixels and those for 210 -> 84 hold 4 or 5 (support = the scale, 1.905 / 2.5): the
// tables carry exactly RS_KH / RS_KV taps per output pixel, shorter windows padded with zero weights (acc + x * 0.0 == acc for the finite,
// non-negative x here: a padded tap is an exact no-op, so are the two / two always-zero taps rounds 1-5 carried at 5 / 7 per pixel and dropped in
// round 6 -- same bits, 20 % / 29 % fewer f64 multiply-adds in the renderer, which is bound by its own VALU stream: profiles/r06_pmc_lockstep_kernels.json)
constexpr int RS_KH = 4, RS_KV = 5;
typedef float f32q __attribute__((ext_vector_type(4)));
struct ResizeLds {
    double kh[84 * RS_KH];
    double kv[84 * RS_KV];
    uint8_t xmin[84];
    uint8_t ymin[84];
    float gray2[256];
};

struct EnvLds {
    ResizeLds R;
    uint8_t ram_prev[128];
    uint8_t ram_cur[128];
    uint16_t off_of_y[212];                 // screen row -> byte offset of its unique row in tmp (slot * 336), premultiplied: one add per tap
    int slot_of_key[192];
    int rep_y[ENV_MAX_ROWS];
    uint8_t img[ENV_MAX_ROWS * 160];        // colour pair (prev<<4 | cur) per pixel of each unique row
    alignas(16) float tmp[ENV_MAX_ROWS * 84];   // horizontally resized unique rows (float32 like PIL's temp image)
    int misc[4];
};

__device__ __forceinline__ int synth_row_class(int y) {
    if (y < 8 || y >= 208) return 0;
    if (y < 16) return 1;
    if (y < 20) return 2;
    if (y < 40) return 3;
    if (y < 64) return 4 + ((63 - y) / 6) * 2 + (y >= 52);
    if (y < 80) return 12;
    return 13 + ((y - 80) >> 2);
}

__device__ __forceinline__ bool synth_player_row(const uint8_t *ram, int y) {
    const int prow = ram[RM_PROW];
    const int py = prow == 0 ? 62 : 48 + 32 * prow;
    const bool blink = ram[RM_FREEZE] > 0 && (ram[RM_FC] & 4);
    return !blink && (unsigned)(y - py) < 16u;
}

// copy the host-built table image (engine.hip: make_lds_tables) into LDS, 8 bytes per thread-step
__device__ inline void synth_load_tables(EnvLds &s, const ResizeLds *__restrict__ T) {
    static_assert(sizeof(ResizeLds) % 8 == 0, "ResizeLds is copied in 8-byte words");
    const unsigned long long *src = (const unsigned long long *)T;
    unsigned long long *dst = (unsigned long long *)&s.R;
    for (int i = threadIdx.x; i < (int)(sizeof(ResizeLds) / 8); i += blockDim.x) dst[i] = src[i];
}

// Render + warp max(prev, cur) and either shift it into the stack (fill == false) or fill all four
// channels with it (fill == true, FrameStack reset).  Called by the whole workgroup (>= 256 threads) after
// synth_load_tables and after ram_prev / ram_cur are in LDS.
// stack_in (optional): where the member's current frame stack is read when the shifted stack goes to another buffer
// (speculative tail: one candidate stack per action); default = in place.
// TAG: a copy of the function per caller class -- TAG 1 is k_env_render's own, so that kernel's register bound (six waves per SIMD = three 512-thread
// workgroups per CU) reaches this code without binding the tail kernels that call TAG 0
template <int TAG = 0>
__device__ inline void synth_observe(EnvLds &s, uint32_t *__restrict__ stack, bool fill, int band = 0, int nbands = 1,
                                     const uint32_t *__restrict__ stack_in = nullptr) {
    if (!stack_in) stack_in = stack;
    // A workgroup produces output rows [yy0, yy1) of the 84 (all of them when nbands == 1; in the tail of a generation a
    // member's frame is split over several workgroups = several CUs).  It needs screen rows [ylo, yhi) only.
    const int tid = threadIdx.x, nthr = blockDim.x;   // 256 threads normally, 1024 when few members remain
    const int yy0 = band * 84 / nbands, yy1 = (band + 1) * 84 / nbands;
    const int ylo = s.R.ymin[yy0], yhi = s.R.ymin[yy1 - 1] + RS_KV;
    const int out0 = yy0 * 84;
    // The vertical pass works on QUADS of output pixels (four neighbours of one output row: 21 quads per row): a quad shares its row's five
    // coefficients and five source-row offsets, reads the four floats of a tap as ONE 16-byte LDS word and moves its four stack words as one
    // 16-byte load and store -- 4 LDS and 21 vector instructions per pixel instead of 16 and 46 (rounds 1-6a: one pixel per thread-step; the
    // pass was 6.5 of a full-width workgroup's 16.1 us, bound by its own instruction stream: tools/render_phase_clock.py).  Per pixel the
    // same five products are added in the same order: same bits.
    constexpr int QPR = 21, VI = 4;                 // quads per output row; quads in flight per thread
    const int nitems = (yy1 - yy0) * QPR;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 oldq[VI];
#pragma unroll
    for (int j = 0; j < VI; j++) oldq[j] = u32x4{0u, 0u, 0u, 0u};
    const u32x4 *stack_in4 = (const u32x4 *)(stack_in + out0);   // (out0 and every row are multiples of 84 = 4 * 21 words: 16-byte aligned)
    u32x4 *stack4 = (u32x4 *)(stack + out0);
    const bool pre = !fill && nthr * 2 >= nitems;   // the whole band's old words fit two quads per thread: asked for before the first barrier
    if (pre) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int it = tid + j * nthr;
            if (it < nitems) oldq[j] = stack_in4[it];
        }
    }
    if (tid < 192) s.slot_of_key[tid] = -1;
    if (tid == 0) s.misc[2] = 0;
    __syncthreads();
    DNE_PHASE(0, 1);
    int key = 0;
    const bool myrow = tid >= ylo && tid < yhi;
    if (myrow) {
        key = synth_row_class(tid) * 4 + (synth_player_row(s.ram_prev, tid) ? 2 : 0) + (synth_player_row(s.ram_cur, tid) ? 1 : 0);
        if (atomicCAS(&s.slot_of_key[key], -1, -2) == -1) {     // first row of this key claims a slot
            int slot = atomicAdd(&s.misc[2], 1);
            if (slot >= ENV_MAX_ROWS) slot = ENV_MAX_ROWS - 1;   // unreachable (<= 75 keys by construction); never write out of bounds
            s.rep_y[slot] = tid;
            s.slot_of_key[key] = slot;
        }
    }
    __syncthreads();
    DNE_PHASE(0, 5);
    if (myrow) s.off_of_y[tid] = (uint16_t)(s.slot_of_key[key] * (84 * 4));
    const int nu = min(s.misc[2], ENV_MAX_ROWS);
    // Row descriptions (round 6): one LANE per (unique row, frame) -- every row of each frame reduced to a background colour + one periodic
    // pattern + one span + the player sprite by the straight-line selects of synth_row_desc, with the row number in a vector register: two
    // waves describe all <= 80 rows of both frames at once (rounds 2-5: one WAVE per row on the scalar unit, 170 scalar instructions per row,
    // 7.5 of a full-width workgroup's 19.3 us -- tools/render_phase_clock.py).  The 13 fields of a description lie in the part of LDS the
    // horizontal pass will fill later (tmp: a barrier lies between), 12 ints per (row, frame): sprite flag and position share one.
    int *desc = (int *)s.tmp;
    static_assert(ENV_MAX_ROWS * 2 * 12 <= ENV_MAX_ROWS * 84, "the row descriptions fit the region they borrow");
    if (tid < 256) {                                    // (the four waves that describe rows unpack the RAM snapshots; the others go straight to the barrier)
        const int f = tid >> 7, u = tid & 127;          // threads 0..127: the previous frame's rows, 128..255: the current frame's
        const PixState pf = synth_pix_uniform(synth_pix_state(f == 0 ? s.ram_prev : s.ram_cur));   // (f is the same in every lane of a wave)
        if (u < nu) {
            const int y = s.rep_y[u];
            DNE_ROW_FIELDS(d_);
            synth_row_desc(pf, y, DNE_ROW_ARGS(d_));
            int4 *o = (int4 *)(desc + (u * 2 + f) * 12);
            o[0] = make_int4(d_bg, d_p_off, d_p_len, d_p_period);
            o[1] = make_int4(d_p_recip, d_p_width, d_p_col, (int)d_p_mask);
            o[2] = make_int4(d_r_lo, d_r_len, d_r_col, d_s_on | (d_s_px << 1));
        }
    }
    __syncthreads();
    DNE_PHASE(0, 6);
    // one wave per unique row: the description is the same in every lane (a broadcast read), the per-pixel work a short branch-free select chain
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = nthr >> 6, lane = tid & 63;
    for (int u = wave; u < nu; u += nwave) {
        const int4 *da = (const int4 *)(desc + u * 24);
        int c0, c1, c2;
        {   // the previous frame's colours first, then the current frame's: one description live at a time (register footprint: three workgroups per CU)
            const int4 a0 = da[0], a1 = da[1], a2 = da[2];
            synth_row_paint3(lane, a0, a1, a2, c0, c1, c2);
            c0 <<= 4; c1 <<= 4; c2 <<= 4;
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const int4 b0 = da[3], b1 = da[4], b2 = da[5];
            int d0, d1, d2;
            synth_row_paint3(lane, b0, b1, b2, d0, d1, d2);
            c0 |= d0; c1 |= d1; c2 |= d2;
        }
        uint8_t *row = s.img + u * 160;
        row[lane] = (uint8_t)c0;
        row[lane + 64] = (uint8_t)c1;
        if (lane < 32) row[lane + 128] = (uint8_t)c2;
    }
    __syncthreads();
    DNE_PHASE(0, 2);
    {   // horizontal pass over the unique rows: a thread keeps ONE output column (its window start and four coefficients in registers) and walks
        // down the unique rows, nthr / 84 rows apart -- no index arithmetic and no coefficient reads per output (rounds 1-6a: output i = tid + k * nthr,
        // a division and four 8-byte LDS reads each).  The same four products added in the same order: same bits.
        const int G = nthr / 84;
        if (tid < G * 84) {
            const int u0 = tid / 84, xx = tid - u0 * 84;
            const int x0 = s.R.xmin[xx];
            const double *k = s.R.kh + xx * RS_KH;
            const double k0 = k[0], k1 = k[1], k2 = k[2], k3 = k[3];
            static_assert(RS_KH == 4, "the horizontal pass is written out for four taps");
            for (int u = u0; u < nu; u += G) {
                const uint8_t *px = s.img + u * 160 + x0;
                double acc = 0.0;
                acc = acc + (double)s.R.gray2[px[0]] * k0;
                acc = acc + (double)s.R.gray2[px[1]] * k1;
                acc = acc + (double)s.R.gray2[px[2]] * k2;
                acc = acc + (double)s.R.gray2[px[3]] * k3;
                s.tmp[u * 84 + xx] = (float)acc;
            }
        }
    }
    __syncthreads();
    DNE_PHASE(0, 3);
    // vertical pass + u8 truncation + stack shift, a quad at a time (above).  The old stack words of a thread's VI quads are asked for together
    // (all of them before the first barrier when two per thread cover the band) so that the global-load latency is paid once per chunk.
    for (int base = 0; base < nitems; base += nthr * VI) {
        if (!pre && !fill) {
#pragma unroll
            for (int j = 0; j < VI; j++) {
                const int it = base + tid + j * nthr;
                if (it < nitems) oldq[j] = stack_in4[it];
            }
        }
#pragma unroll
        for (int j = 0; j < VI; j++) {
            const int it = base + tid + j * nthr;
            if (it < nitems) {
                const int r = (int)(__umul24((unsigned)it, 3121u) >> 16), q = it - r * QPR, yy = yy0 + r;   // it / 21, exact below 4096 (it < 1764), on the full-rate multiplier
                const uint16_t *sl = s.off_of_y + s.R.ymin[yy];
                const double *k = s.R.kv + __umul24((unsigned)yy, (unsigned)RS_KV);
                const char *col = (const char *)s.tmp + q * 16;
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int t = 0; t < RS_KV; t++) {
                    const f32q v = *(const f32q *)(col + sl[t]);
                    const double kt = k[t];
                    a0 = a0 + (double)v[0] * kt;
                    a1 = a1 + (double)v[1] * kt;
                    a2 = a2 + (double)v[2] * kt;
                    a3 = a3 + (double)v[3] * kt;
                }
                const u32x4 pix = {(uint32_t)(uint8_t)(float)a0, (uint32_t)(uint8_t)(float)a1, (uint32_t)(uint8_t)(float)a2, (uint32_t)(uint8_t)(float)a3};
                stack4[it] = fill ? pix * 0x01010101u : ((oldq[j] >> 8) | (pix << 24));
            }
        }
    }
}

}  // namespace dne
