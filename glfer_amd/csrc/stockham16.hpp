// stockham16.hpp -- pieces shared by the 16-points-per-lane Stockham kernels (spectro16.hip: the
// packed-pair N-point form; spectro16h.hip: the real-input N/2-point form): the radix schedule,
// the range-checked sample gather, the frame barrier and the pass loop with its LDS exchange.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft_inreg.hpp"
#include "spectro_params.h"

// Timeline stamps for tools/stampbench only: GLFER_STAMP(id) records the shader clock of one chosen
// wave at a phase boundary.  Expands to nothing in the product build.
#ifndef GLFER_STAMP
#define GLFER_STAMP(id)
#endif
// 1: the barrier that frees the exchange buffer sits right after an exchange's reads; 0: in front
// of the next exchange's writes
// inter-pass twiddles folded into the first butterfly stage (fft_inreg.hpp: dit_tw)
#ifndef GLFER16_FOLD_TW
#define GLFER16_FOLD_TW 1
#endif
#ifndef GLFER16_BARRIER_AFTER_READS
#define GLFER16_BARRIER_AFTER_READS 1
#endif

namespace glfer {

typedef float v2f32 __attribute__((ext_vector_type(2)));

// Radix schedule for N = 2^LOGN with 16 points per lane.
template <int LOGN>
struct Plan16 {
  static constexpr int N = 1 << LOGN;
  static constexpr int T = N / 16;                                   // lanes per frame
  static constexpr int NPASS = LOGN <= 8 ? 2 : (LOGN <= 12 ? 3 : 4);
  static constexpr int radix(int i) {
    if (i < 2) return 16;
    if (NPASS == 3) return N / 256;
    return i == 2 ? 16 : N / 4096;
  }
  static constexpr int ls(int i) {                                    // product of earlier radices
    int l = 1;
    for (int j = 0; j < i; j++) l *= radix(j);
    return l;
  }
  static constexpr int tw_offset(int i) {                             // first twiddle slot of pass i
    int o = 0;
    for (int j = 1; j < i; j++) o += (16 / radix(j)) * (radix(j) - 1);
    return o;
  }
  static constexpr int NTW = tw_offset(NPASS);                        // twiddles per lane
};


// sample formats: wav_fmt.c:104-117
template <int FMT>
__device__ __forceinline__ float cvt_sample(const void *ubase, unsigned elem_off) {
  if constexpr (FMT == GLFER_FMT_F32) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(ubase) + (size_t)(elem_off * 4u));
  } else if constexpr (FMT == GLFER_FMT_S16) {
    return (float)*reinterpret_cast<const short *>(reinterpret_cast<const char *>(ubase) + (size_t)(elem_off * 2u)) / 32768.0f;
  } else {
    return ((float)*(reinterpret_cast<const unsigned char *>(ubase) + (size_t)elem_off) - 128.0f) / 128.0f;
  }
}

// Range-checked buffer load of one sample (raw buffer: out-of-range offsets read 0): one shared
// VGPR byte offset + an SGPR/immediate offset, so gathering a frame costs no address VALU.
#ifndef GLFER_X_LOAD_AUX
#define GLFER_X_LOAD_AUX 0        /* cache policy of the sample loads (A/B builds: 2 = non-temporal) */
#endif
template <int FMT>
__device__ __forceinline__ float buf_sample(__amdgpu_buffer_rsrc_t rsrc, unsigned voff_bytes, unsigned soff_bytes) {
  if constexpr (FMT == GLFER_FMT_F32) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff_bytes, soff_bytes, GLFER_X_LOAD_AUX));
  } else if constexpr (FMT == GLFER_FMT_S16) {
    return (float)(short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, voff_bytes, soff_bytes, GLFER_X_LOAD_AUX) / 32768.0f;
  } else {
    return ((float)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(rsrc, voff_bytes, soff_bytes, GLFER_X_LOAD_AUX) - 128.0f) / 128.0f;
  }
}

// The limiter of prepare_audio (fft.c:151-156): `ftmp = log(fabs(y))` is a double logarithm rounded to the
// reference's FLOAT ftmp, `exp(ftmp * 0.1)` a double product and a double exponential rounded to float on the
// store into inbuf_fft.  Done exactly so (round 4; the fast float intrinsics were 2e-4 of the row maximum away:
// |y|^0.1 flattens the frame, so every sample's rounding shows).  A cold path: the limiter is off by default
// (glfer.c:241) and the estimator kernels reach it through their general form only.
__device__ __forceinline__ float limiter_value(float y) {
  const float ftmp = (float)log((double)fabsf(y));
  const float mag = (float)exp((double)ftmp * 0.1);
  return y > 0.0f ? mag : -mag;
}

// Workgroup w of a launch runs on XCD (w mod 8), each XCD with its own L2.  Overlapped frames share
// samples with their neighbours, so neighbouring frame blocks should share an L2: logical block
// index = the XCD's contiguous slice of the grid (gridDim.x a multiple of 8; identity otherwise).
__device__ __forceinline__ unsigned xcd_block_index() {
  const unsigned w = blockIdx.x, g = gridDim.x;
  return (g & 7u) ? w : (w & 7u) * (g >> 3) + (w >> 3);
}

// GLFER_ABL (tools/xbench timing ablations only; results are wrong): bit 0 = no workgroup barriers,
// bit 1 = no exchange writes, bit 2 = no exchange reads
#ifndef GLFER_ABL
#define GLFER_ABL 0
#endif
template <int T>
__device__ __forceinline__ void frame_sync() {
  if constexpr (T > 64 && !(GLFER_ABL & 1)) {
    __syncthreads();
  } else {                       // the frame lives in one wave: LDS ops of a wave are in order
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <int LOGN>
struct Launch16 {
  static constexpr int T = Plan16<LOGN>::T;
  static constexpr int FPB = T >= 256 ? 1 : 256 / T;          // frames per block
  static constexpr int BLOCK = T * FPB;
  static constexpr int PADN = Plan16<LOGN>::N + Plan16<LOGN>::N / 16;   // covers both exchange layouts
  static constexpr int LDS_WORDS = FPB * PADN + 16 * 17;       // + pass-1 twiddle table [16][17] (padded rows)
};

// Padded exchange layout: logical index a lives at entry a + (a >> shift).  Exchange 0 is written
// 16 consecutive entries per lane (a = 16 t + q): +1 per 16 makes its 16-lane ds_write_b64 groups
// conflict free, and leaves one 2-way conflict in every 32-lane ds_read_b64 group (33 entries).
// Later exchanges are written one entry per lane, 16 consecutive lanes to 16 consecutive entries:
// +1 per 32 is conflict free for those writes AND for the reads (32 consecutive entries).
// (tools/ldsbench2: reading with the 2-way conflict runs at 108 B/clk/CU, half the pipe's rate.)
// xpad_offset(i, d): entry distance for a logical distance d that is a compile-time multiple of 16
// added to a base whose low bits cannot carry into it (asserted by the callers' index algebra).
__host__ __device__ constexpr int xpad_shift(int exchange) { return exchange == 0 ? 4 : 5; }
__host__ __device__ constexpr int xpad_offset(int exchange, int d) { return d + (d >> xpad_shift(exchange)); }

// Exchange 0 in ROW layout (GLFER16_X0_ROWS, T >= 32): output q of producer lane t goes to row q,
// column t (row stride T + 2 entries), so a 16-lane ds_write_b64 group writes 16 consecutive
// entries; consumer lane 16 u + k reads row k at columns u + 16 m.  (T + 2) = 2 mod 32 puts the
// 32 (k, u) combinations of a read group on 32 distinct bank pairs: conflict free BOTH ways.
// The 16 reads of a lane sit T/2 bytes apart, which LLVM's load/store optimiser would fuse into
// ds_read2_b64 -- half the rate of ds_read_b64 (MI355X_MICROARCH.md, LDS table) -- so they are
// issued as inline asm; the explicit s_waitcnt that makes their results visible costs nothing,
// because the buffer-release barrier follows the reads anyway.
#ifndef GLFER16_X0_ROWS
#define GLFER16_X0_ROWS 1
#endif
template <int BYTE_OFFSET>
__device__ __forceinline__ void lds_read_b64_asm(unsigned addr, v2f32 &out) {
  // "memory": the compiler must not move this read across the LDS stores it cannot see it depends on
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(out) : "v"(addr), "n"(BYTE_OFFSET) : "memory");
}
// v[m] = base[m * STRIDE] (STRIDE in entries), m = 0..15
template <int STRIDE>
__device__ __forceinline__ void lds_read16_strided(const v2f32 *base, v2f32 (&v)[16]) {
  static_assert(15 * STRIDE * 8 < 65536, "ds offset field");
  const unsigned addr = (unsigned)(__SIZE_TYPE__)(const __attribute__((address_space(3))) char *)base;
  static_for<0, 16>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    lds_read_b64_asm<m * STRIDE * 8>(addr, v[m]);
  });
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                 "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
               :
               : "memory");
}

// The same reads issued WITHOUT the wait, and the wait on its own: lds_wait16<K>(v) returns when at most K LDS
// operations issued after v's reads are still outstanding (LDS operations complete in order), and ties v's
// registers to that point for the compiler.  Between the issue and the wait the compiler must have no reason to
// touch v (callers keep it out of every expression until the wait).
template <int STRIDE>
__device__ __forceinline__ void lds_issue16_strided(const v2f32 *base, v2f32 (&v)[16]) {
  static_assert(15 * STRIDE * 8 < 65536, "ds offset field");
  const unsigned addr = (unsigned)(__SIZE_TYPE__)(const __attribute__((address_space(3))) char *)base;
  static_for<0, 16>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    lds_read_b64_asm<m * STRIDE * 8>(addr, v[m]);
  });
}
template <int K>
__device__ __forceinline__ void lds_wait16(v2f32 (&v)[16]) {
  asm volatile("s_waitcnt lgkmcnt(%16)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                 "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
               : "n"(K)
               : "memory");
}

// The lane's pass-1 twiddles as the passes take them (tw1[q], q = 1..15, q a compile-time constant): a row of the
// shared LDS table, or -- REGS -- the lane's own 30 registers (15 fewer ds_read_b64 per transform; worth 1-3 % where
// the kernel's occupancy survives the registers: profiles/r03_y_tw1_regs.txt, r03_h_tw1_regs.txt).
template <bool REGS>
struct Tw1Source {
  v2f32 r[REGS ? 16 : 1];
  const v2f32 *row;
  __device__ __forceinline__ void init(const v2f32 *tw1row) {
    row = tw1row;
    if constexpr (REGS) {
#pragma unroll
      for (int q = 0; q < 16; q++) r[q] = tw1row[q];
    }
  }
  __device__ __forceinline__ v2f32 operator[](int q) const {
    if constexpr (REGS) return r[q];
    else return row[q];
  }
};

// pass I's input twiddles by register index, for dit_tw: none in pass 0 and on a sub-transform's element 0
template <class C, int I, int NT, class Tw1>
struct PassTwiddles {
  const Tw1 &tw1row;
  const float (&twr)[NT];
  const float (&twi)[NT];
  template <int M>
  __device__ __forceinline__ auto operator()(std::integral_constant<int, M>) const {
    constexpr int R = C::radix(I), B = 16 / R, b = M % B, q = M / B;
    if constexpr (I == 0 || q == 0) {
      return NoTwiddle{};
    } else if constexpr (I == 1) {
      return tw1row[M];
    } else {
      constexpr int e = C::tw_offset(I) - 15 + b * (R - 1) + (q - 1);
      return v2f32{twr[e], twi[e]};
    }
  }
};

// The Stockham passes of one complex 2^LOGM-point transform held 16 points per lane
// (lane t of T = 2^LOGM/16: points t + T*m on entry; on exit register b + B*brev(q',R) holds
// bin t + T*(b + B*q'), R = last radix, B = 16/R).  xb: this frame's exchange buffer
// (M + M/16 entries), tw1row: the lane's row of the shared pass-1 table, twr/twi: the
// lane's later-pass twiddles.  after_first_write() runs between the first exchange's writes
// and its barrier: the place where the next round's global loads are issued.
template <int LOGM, int NT, class Tw1, class Hook>
__device__ __forceinline__ void stockham16_passes(float (&zr)[16], float (&zi)[16], v2f32 *xb, unsigned t,
                                                  const Tw1 &tw1row, const float (&twr)[NT],
                                                  const float (&twi)[NT], Hook &&after_first_write) {
  using C = Plan16<LOGM>;
  constexpr int T = C::T, NPASS = C::NPASS, TW1 = 15;
  static_for<0, NPASS>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int R = C::radix(i), Ls = C::ls(i), B = 16 / R;
    constexpr bool kFold = GLFER16_FOLD_TW != 0 && i > 0;
    const PassTwiddles<C, i, NT, Tw1> twf{tw1row, twr, twi};
    if constexpr (kFold) {
    } else if constexpr (i == 1) {
      static_for<1, 16>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const v2f32 w = tw1row[q];
        const float a = zr[q], c = zi[q];
        zr[q] = __builtin_fmaf(a, w.x, -c * w.y);
        zi[q] = __builtin_fmaf(a, w.y, c * w.x);
      });
    } else if constexpr (i > 1) {
      static_for<0, B>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        static_for<1, R>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          constexpr int e = C::tw_offset(i) - TW1 + b * (R - 1) + (q - 1);
          constexpr int m = b + B * q;
          const float a = zr[m], c = zi[m];
          zr[m] = __builtin_fmaf(a, twr[e], -c * twi[e]);
          zi[m] = __builtin_fmaf(a, twi[e], c * twr[e]);
        });
      });
    }
    constexpr bool kEmit = GLFER16_BARRIER_AFTER_READS != 0 && i < NPASS - 1 && B == 1;
    if constexpr (kEmit) {
      // nothing fences the exchange's writes off from the butterflies that produce them (the
      // barrier sits after the previous reads): each output goes to LDS as soon as it exists
      const int k = (int)t & (Ls - 1);
      const int a0 = ((int)t - k) * R + k;
      constexpr bool kRows = GLFER16_X0_ROWS != 0 && i == 0 && T >= 32;
      v2f32 *wbase = kRows ? xb + t : xb + a0 + (a0 >> xpad_shift(i));
      if constexpr (kFold) dit_head_tw<R, 1, 0, 16>(zr, zi, twf);
      else dit_head<R, 1, 0, 16>(zr, zi);
      dit_tail<R, 1, 0, 16>(zr, zi, [&](auto qc, auto rc) {
        constexpr int q = decltype(qc)::value, reg = decltype(rc)::value;
        if constexpr (!(GLFER_ABL & 2)) wbase[kRows ? q * (T + 2) : xpad_offset(i, q * Ls)] = v2f32{zr[reg], zi[reg]};
      });
    } else {
      static_for<0, B>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        if constexpr (kFold) dit_tw<R, B, b, 16>(zr, zi, twf);
        else dit<R, B, b, 16>(zr, zi);
      });
    }
    GLFER_STAMP(4 * i + 1);                // pass i butterflies done
    if constexpr (i < NPASS - 1) {
      if constexpr (GLFER16_BARRIER_AFTER_READS == 0) frame_sync<T>();   // everyone has finished reading the previous exchange
      GLFER_STAMP(4 * i + 2);              // through the pre-write barrier
      if constexpr (!kEmit) static_for<0, B>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        const int j = (int)t + T * b;
        const int k = j & (Ls - 1);
        const int a0 = (j - k) * R + k;
        v2f32 *wbase = xb + a0 + (a0 >> xpad_shift(i));
        static_for<0, R>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          constexpr int src = b + B * brev(q, R);
          wbase[xpad_offset(i, q * Ls)] = v2f32{zr[src], zi[src]};
        });
      });
      if constexpr (i == 0) {
        after_first_write();
        __builtin_amdgcn_sched_barrier(0);
      }
      GLFER_STAMP(4 * i + 3);              // writes (and the hook's loads) issued
      frame_sync<T>();
      GLFER_STAMP(4 * i + 4);              // through the post-write barrier
      if constexpr (GLFER_ABL & 4) {
      } else if constexpr (GLFER16_X0_ROWS != 0 && GLFER16_BARRIER_AFTER_READS != 0 && i == 0 && T >= 32 && B == 1) {
        v2f32 v[16];
        lds_read16_strided<T / 16>(xb + (t & 15) * (T + 2) + (t >> 4), v);
#pragma unroll
        for (int m = 0; m < 16; m++) {
          zr[m] = v[m].x;
          zi[m] = v[m].y;
        }
      } else {
        const v2f32 *rbase = xb + t + (t >> xpad_shift(i));
        // entry distance between a lane's reads: xpad_offset(i, m*T) = m * (T + T/32) when T is a
        // multiple of 32 -- 16 plain ds_read_b64 with immediate offsets where they fit the 16-bit
        // field (the compiler's own choice is ds_read2_b64 / ds_read2st64_b64: half the rate)
        constexpr int STRIDE = xpad_offset(i, T);
        if constexpr (GLFER16_X0_ROWS != 0 && T % 32 == 0 && i > 0 && 15 * STRIDE * 8 < 65536 && xpad_offset(i, 15 * T) == 15 * STRIDE) {
          v2f32 v[16];
          lds_read16_strided<STRIDE>(rbase, v);
#pragma unroll
          for (int m = 0; m < 16; m++) {
            zr[m] = v[m].x;
            zi[m] = v[m].y;
          }
        } else {
          static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const v2f32 v = rbase[xpad_offset(i, m * T)];
            zr[m] = v.x;
            zi[m] = v.y;
          });
        }
      }
      // With the barrier here (the reads have landed: the barrier waits for lgkmcnt(0)) instead
      // of in front of the next writes, those writes are not fenced off from the butterflies
      // that produce them and can be issued as their data becomes ready.  The buffer is then
      // free for ANY writer after this point, including the caller's fold / mirror steps.
      if constexpr (GLFER16_BARRIER_AFTER_READS != 0) frame_sync<T>();
    }
  });
}

// Two transforms (two frames, two exchange buffers) interleaved in one wavefront: in every phase
// between barriers the wavefront first does stream A's butterflies -- its exchange writes leave
// from inside the last butterfly stage -- and then stream B's, so A's writes drain through the LDS
// pipe under B's arithmetic, and each barrier serves both transforms (4 per pair of transforms
// instead of 8).  Same arithmetic per stream as stockham16_passes; requires the
// barrier-after-reads scheme and radix-16 passes in front of every exchange.
// tw1row: the lane's pass-1 twiddles, tw1row[q], q = 1..15 -- a row of the shared LDS table (const v2f32 *) or the
// lane's own registers (const v2f32 (&)[16]: GLFER16Y_TW1_REGS)
template <int LOGM, int NT, class Tw1, class Hook>
__device__ __forceinline__ void stockham16_passes2(float (&zrA)[16], float (&ziA)[16], v2f32 *xbA, float (&zrB)[16],
                                                   float (&ziB)[16], v2f32 *xbB, unsigned t, const Tw1 &tw1row,
                                                   const float (&twr)[NT], const float (&twi)[NT],
                                                   Hook &&after_first_write) {
  using C = Plan16<LOGM>;
  constexpr int T = C::T, NPASS = C::NPASS, TW1 = 15;
  static_assert(GLFER16_BARRIER_AFTER_READS != 0, "stockham16_passes2 relies on the barrier-after-reads scheme");
  static_for<0, NPASS>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int R = C::radix(i), Ls = C::ls(i), B = 16 / R;
    constexpr bool kFold = GLFER16_FOLD_TW != 0 && i > 0;
    const PassTwiddles<C, i, NT, Tw1> twf{tw1row, twr, twi};
    auto compute = [&](float (&zr)[16], float (&zi)[16], v2f32 *xb) {
      if constexpr (kFold) {
      } else if constexpr (i == 1) {
        static_for<1, 16>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          const v2f32 w = tw1row[q];
          const float a = zr[q], c = zi[q];
          zr[q] = __builtin_fmaf(a, w.x, -c * w.y);
          zi[q] = __builtin_fmaf(a, w.y, c * w.x);
        });
      } else if constexpr (i > 1) {
        static_for<0, B>([&](auto bc) {
          constexpr int b = decltype(bc)::value;
          static_for<1, R>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int e = C::tw_offset(i) - TW1 + b * (R - 1) + (q - 1);
            constexpr int m = b + B * q;
            const float a = zr[m], c = zi[m];
            zr[m] = __builtin_fmaf(a, twr[e], -c * twi[e]);
            zi[m] = __builtin_fmaf(a, twi[e], c * twr[e]);
          });
        });
      }
      if constexpr (i < NPASS - 1) {
        static_assert(B == 1, "an exchange follows radix-16 passes only");
        const int k = (int)t & (Ls - 1);
        const int a0 = ((int)t - k) * R + k;
        constexpr bool kRows = GLFER16_X0_ROWS != 0 && i == 0 && T >= 32;
        v2f32 *wbase = kRows ? xb + t : xb + a0 + (a0 >> xpad_shift(i));
        if constexpr (kFold) dit_head_tw<R, 1, 0, 16>(zr, zi, twf);
        else dit_head<R, 1, 0, 16>(zr, zi);
        dit_tail<R, 1, 0, 16>(zr, zi, [&](auto qc, auto rc) {
          constexpr int q = decltype(qc)::value, reg = decltype(rc)::value;
          if constexpr (!(GLFER_ABL & 2)) wbase[kRows ? q * (T + 2) : xpad_offset(i, q * Ls)] = v2f32{zr[reg], zi[reg]};
        });
      } else {
        static_for<0, B>([&](auto bc) {
          constexpr int b = decltype(bc)::value;
          if constexpr (kFold) dit_tw<R, B, b, 16>(zr, zi, twf);
          else dit<R, B, b, 16>(zr, zi);
        });
      }
    };
    compute(zrA, ziA, xbA);
    GLFER_STAMP(4 * i + 1);                  // stream A: pass i done, its writes issued
    __builtin_amdgcn_sched_barrier(0);
    compute(zrB, ziB, xbB);
    if constexpr (i < NPASS - 1) {
      if constexpr (i == 0) after_first_write();
      __builtin_amdgcn_sched_barrier(0);
      GLFER_STAMP(4 * i + 2);                // stream B: pass i done, its writes (and the hook's loads) issued
      frame_sync<T>();                       // both streams' writes are in LDS
      GLFER_STAMP(4 * i + 3);                // through the post-write barrier
      if constexpr (GLFER_ABL & 4) {
      } else if constexpr (GLFER16_X0_ROWS != 0 && i == 0 && T >= 32) {
        v2f32 va[16], vb[16];
        const int roff = (int)(t & 15) * (T + 2) + (int)(t >> 4);
        lds_read16_strided<T / 16>(xbA + roff, va);
        lds_read16_strided<T / 16>(xbB + roff, vb);
#pragma unroll
        for (int m = 0; m < 16; m++) {
          zrA[m] = va[m].x;
          ziA[m] = va[m].y;
          zrB[m] = vb[m].x;
          ziB[m] = vb[m].y;
        }
      } else {
        const v2f32 *ra = xbA + t + (t >> xpad_shift(i)), *rb = xbB + t + (t >> xpad_shift(i));
        constexpr int STRIDE = xpad_offset(i, T);
        if constexpr (GLFER16_X0_ROWS != 0 && T % 32 == 0 && i > 0 && 15 * STRIDE * 8 < 65536 && xpad_offset(i, 15 * T) == 15 * STRIDE) {
          v2f32 va[16], vb[16];                // plain ds_read_b64, immediate offsets (see stockham16_passes)
          lds_read16_strided<STRIDE>(ra, va);
          lds_read16_strided<STRIDE>(rb, vb);
#pragma unroll
          for (int m = 0; m < 16; m++) {
            zrA[m] = va[m].x;
            ziA[m] = va[m].y;
            zrB[m] = vb[m].x;
            ziB[m] = vb[m].y;
          }
        } else {
          static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const v2f32 v = ra[xpad_offset(i, m * T)];
            zrA[m] = v.x;
            ziA[m] = v.y;
          });
          static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const v2f32 v = rb[xpad_offset(i, m * T)];
            zrB[m] = v.x;
            ziB[m] = v.y;
          });
        }
      }
      frame_sync<T>();                       // both buffers read: free for the next writes
      GLFER_STAMP(4 * i + 4);                // both streams' reads landed, through the barrier
    }
  });
}

// stockham16_passes2 with stream B's exchange reads and the buffer-release barrier moved under stream A's
// arithmetic (round 3).  After an exchange's writes are in LDS: both streams' reads are ISSUED; A's sixteen are
// waited for alone and A runs its twiddles and all butterfly stages but the last while B's land; then the barrier
// that frees the buffers (every wavefront reaches it with arithmetic done, so the skew between them is absorbed
// there instead of stalling the reads), A's last stage with its exchange writes, and B's whole pass.  Per stream
// the operations and their order are those of stockham16_passes2: bit-identical results.  Radix-16 passes only.
template <int LOGM, int NT, class Tw1, class Hook>
__device__ __forceinline__ void stockham16_passes2s(float (&zrA)[16], float (&ziA)[16], v2f32 *xbA, float (&zrB)[16],
                                                    float (&ziB)[16], v2f32 *xbB, unsigned t, const Tw1 &tw1row,
                                                    const float (&twr)[NT], const float (&twi)[NT],
                                                    Hook &&after_first_write) {
  using C = Plan16<LOGM>;
  constexpr int T = C::T, NPASS = C::NPASS, TW1 = 15;
  static_assert(GLFER16_BARRIER_AFTER_READS != 0 && GLFER16_X0_ROWS != 0 && T >= 32 && T % 32 == 0 && !(GLFER_ABL & 7), "the product configuration");
  static_for<0, NPASS>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int R = C::radix(i), Ls = C::ls(i);
    static_assert(R == 16, "radix-16 passes");
    constexpr bool kFold = GLFER16_FOLD_TW != 0 && i > 0;
    const PassTwiddles<C, i, NT, Tw1> twf{tw1row, twr, twi};
    auto twiddle = [&](float (&zr)[16], float (&zi)[16]) {
      if constexpr (kFold) {
        dit_head_tw<R, 1, 0, 16>(zr, zi, twf);
      } else if constexpr (i == 1) {
        static_for<1, 16>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          const v2f32 w = tw1row[q];
          const float a = zr[q], c = zi[q];
          zr[q] = __builtin_fmaf(a, w.x, -c * w.y);
          zi[q] = __builtin_fmaf(a, w.y, c * w.x);
        });
      } else if constexpr (i > 1) {
        static_for<1, R>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          constexpr int e = C::tw_offset(i) - TW1 + (q - 1);
          const float a = zr[q], c = zi[q];
          zr[q] = __builtin_fmaf(a, twr[e], -c * twi[e]);
          zi[q] = __builtin_fmaf(a, twi[e], c * twr[e]);
        });
      }
      if constexpr (!kFold) dit_head<R, 1, 0, 16>(zr, zi);
    };
    // the last stage: with the exchange writes (a pass that an exchange follows) or plain
    auto tail = [&](float (&zr)[16], float (&zi)[16], v2f32 *xb) {
      if constexpr (i < NPASS - 1) {
        const int k = (int)t & (Ls - 1);
        const int a0 = ((int)t - k) * R + k;
        constexpr bool kRows = i == 0;
        v2f32 *wbase = kRows ? xb + t : xb + a0 + (a0 >> xpad_shift(i));
        dit_tail<R, 1, 0, 16>(zr, zi, [&](auto qc, auto rc) {
          constexpr int q = decltype(qc)::value, reg = decltype(rc)::value;
          wbase[kRows ? q * (T + 2) : xpad_offset(i, q * Ls)] = v2f32{zr[reg], zi[reg]};
        });
      } else {
        dit_tail<R, 1, 0, 16>(zr, zi, [](auto, auto) {});
      }
    };
    if constexpr (i == 0) {
      dit_head<R, 1, 0, 16>(zrA, ziA);
      tail(zrA, ziA, xbA);
      __builtin_amdgcn_sched_barrier(0);
      dit_head<R, 1, 0, 16>(zrB, ziB);
      tail(zrB, ziB, xbB);
      after_first_write();
      __builtin_amdgcn_sched_barrier(0);
      frame_sync<T>();                         // both streams' writes are in LDS
    } else {
      // the reads of exchange i - 1, both streams, issued back to back
      v2f32 va[16], vb[16];
      if constexpr (i == 1) {
        const int roff = (int)(t & 15) * (T + 2) + (int)(t >> 4);
        lds_issue16_strided<T / 16>(xbA + roff, va);
        lds_issue16_strided<T / 16>(xbB + roff, vb);
      } else {
        constexpr int STRIDE = xpad_offset(i - 1, T);
        static_assert(15 * STRIDE * 8 < 65536 && xpad_offset(i - 1, 15 * T) == 15 * STRIDE, "one stride");
        lds_issue16_strided<STRIDE>(xbA + t + (t >> xpad_shift(i - 1)), va);
        lds_issue16_strided<STRIDE>(xbB + t + (t >> xpad_shift(i - 1)), vb);
      }
      lds_wait16<15>(va);                      // (lgkmcnt has four bits) at most 15 of the 32 outstanding: A's sixteen have landed
#pragma unroll
      for (int m = 0; m < 16; m++) {
        zrA[m] = va[m].x;
        ziA[m] = va[m].y;
      }
      twiddle(zrA, ziA);                       // and every butterfly stage but the last
      lds_wait16<0>(vb);
#pragma unroll
      for (int m = 0; m < 16; m++) {
        zrB[m] = vb[m].x;
        ziB[m] = vb[m].y;
      }
      frame_sync<T>();                         // both buffers read: free for the writes that follow
      tail(zrA, ziA, xbA);
      __builtin_amdgcn_sched_barrier(0);
      twiddle(zrB, ziB);
      tail(zrB, ziB, xbB);
      if constexpr (i < NPASS - 1) {
        __builtin_amdgcn_sched_barrier(0);
        frame_sync<T>();                       // both streams' writes are in LDS
      }
    }
  });
}

// One stream, the buffer-release barrier moved in front of the last butterfly stage (the single-stream counterpart of
// stockham16_passes2s: the reads cannot hide under anything, but every wavefront reaches the barrier with its arithmetic
// done).  Same operations in the same order as stockham16_passes.  Radix-16 passes only.
template <int LOGM, int NT, class Tw1, class Hook>
__device__ __forceinline__ void stockham16_passes1s(float (&zr)[16], float (&zi)[16], v2f32 *xb, unsigned t, const Tw1 &tw1row,
                                                    const float (&twr)[NT], const float (&twi)[NT], Hook &&after_first_write) {
  using C = Plan16<LOGM>;
  constexpr int T = C::T, NPASS = C::NPASS, TW1 = 15;
  static_assert(GLFER16_BARRIER_AFTER_READS != 0 && GLFER16_X0_ROWS != 0 && T >= 32 && T % 32 == 0 && !(GLFER_ABL & 7), "the product configuration");
  static_for<0, NPASS>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int R = C::radix(i), Ls = C::ls(i);
    static_assert(R == 16, "radix-16 passes");
    auto tail = [&] {
      if constexpr (i < NPASS - 1) {
        const int k = (int)t & (Ls - 1);
        const int a0 = ((int)t - k) * R + k;
        constexpr bool kRows = i == 0;
        v2f32 *wbase = kRows ? xb + t : xb + a0 + (a0 >> xpad_shift(i));
        dit_tail<R, 1, 0, 16>(zr, zi, [&](auto qc, auto rc) {
          constexpr int q = decltype(qc)::value, reg = decltype(rc)::value;
          wbase[kRows ? q * (T + 2) : xpad_offset(i, q * Ls)] = v2f32{zr[reg], zi[reg]};
        });
      } else {
        dit_tail<R, 1, 0, 16>(zr, zi, [](auto, auto) {});
      }
    };
    if constexpr (i == 0) {
      dit_head<R, 1, 0, 16>(zr, zi);
      tail();
      after_first_write();
      __builtin_amdgcn_sched_barrier(0);
      frame_sync<T>();                         // the writes are in LDS
    } else {
      v2f32 v[16];
      if constexpr (i == 1) {
        lds_read16_strided<T / 16>(xb + (t & 15) * (T + 2) + (t >> 4), v);
      } else {
        constexpr int STRIDE = xpad_offset(i - 1, T);
        static_assert(15 * STRIDE * 8 < 65536 && xpad_offset(i - 1, 15 * T) == 15 * STRIDE, "one stride");
        lds_read16_strided<STRIDE>(xb + t + (t >> xpad_shift(i - 1)), v);
      }
#pragma unroll
      for (int m = 0; m < 16; m++) {
        zr[m] = v[m].x;
        zi[m] = v[m].y;
      }
      if constexpr (GLFER16_FOLD_TW != 0) {
        const PassTwiddles<C, i, NT, Tw1> twf{tw1row, twr, twi};
        dit_head_tw<R, 1, 0, 16>(zr, zi, twf);
      } else {
        if constexpr (i == 1) {
          static_for<1, 16>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const v2f32 w = tw1row[q];
            const float a = zr[q], c = zi[q];
            zr[q] = __builtin_fmaf(a, w.x, -c * w.y);
            zi[q] = __builtin_fmaf(a, w.y, c * w.x);
          });
        } else {
          static_for<1, R>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int e = C::tw_offset(i) - TW1 + (q - 1);
            const float a = zr[q], c = zi[q];
            zr[q] = __builtin_fmaf(a, twr[e], -c * twi[e]);
            zi[q] = __builtin_fmaf(a, twi[e], c * twr[e]);
          });
        }
        dit_head<R, 1, 0, 16>(zr, zi);
      }
      frame_sync<T>();                         // the buffer is read: free for the writes that follow (and for the caller after the last pass)
      tail();
      if constexpr (i < NPASS - 1) {
        __builtin_amdgcn_sched_barrier(0);
        frame_sync<T>();                       // the writes are in LDS
      }
    }
  });
}

}  // namespace glfer
