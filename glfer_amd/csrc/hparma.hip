// hparma.hip -- batched HP-ARMA estimator (BASELINE config 5), one wavefront per frame.
//
// Replaces hparma_do (hparma.c:74-157) + compute_svd (util.c:261-386) for a batch of frames:
//   autocorrelation of the (unwindowed) frame for lags 0..t-1, the t x (p_e+1) "Toeplitz" matrix
//   with the reference's row-0 overflow reproduced through a host-built lag map, one-sided
//   Jacobi SVD in the reference's cyclic column order, rank from the cumulative sigma^2, AR
//   vector from the noise subspace, |A(f)|^2/N on N/2+1 bins and its reciprocal below Nyquist.
// Compute bound by the vector instructions it issues (most of them double precision at 4 clocks each: 230 k a frame at BASELINE
// config 5, 0.81 of the SIMD clocks -- profiles/r04_hparma_issue.json), not HBM-bound: frames are independent, seven of them are in
// flight per CU (22 KB of LDS each) and take their successors from a queue.
// The matrix lives in LDS column-major (a column pair is two conflict-free strided reads per
// lane), inner products are double as in the reference, the three sums of a rotation are reduced
// across the wave with DPP, and every lane repeats the scalar part so all branches are uniform.
// Compiled with -ffp-contract=off: a*c + b*s must round as the reference's two multiplies + add.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "spectro_params.h"

namespace glfer {
hipError_t allow_dynamic_lds(const void *kernel, size_t bytes);   // plan.h / glfer_hip.cpp: once per device, kernel and size class
hipError_t scratch_malloc(void **p, size_t bytes, hipStream_t st);   // (stream-ordered scratch of the library)
void scratch_free(void *q, hipStream_t st);
}

namespace glfer {

// sum of v over the 64 lanes, result in every lane
__device__ __forceinline__ double wave_sum(double v) {
  auto dpp = [](double x, auto ctrl) -> double {
    constexpr int C = decltype(ctrl)::value;
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, C, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), C, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror: every lane holds its row's sum
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  double tot = 0.0;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 16 * r);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 16 * r);
    tot += __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  }
  return tot;
}

// The three sums of a rotation at once (round 3).  Three separate wave_sum()s are 24 double-precision additions (4 clocks each)
// and 48 cross-lane moves; here the lanes SHARE the work: after the first exchange (lane ^ 1) the even lanes carry a and b, the
// odd lanes c; after the second (lane ^ 2) lane q of every quad carries one value -- q = 0: a, 1: c, 2: b, 3: nothing -- and from
// there one tree (row_ror 4 and 8, then the rows by v_permlane16_swap / v_permlane32_swap) sums all three: 7 additions.  Lanes 0, 1
// and 2 hand every lane the three totals (v_readlane: the same bits everywhere).  The order of the additions is fixed (a tree over the lanes, as before).
__device__ __forceinline__ void wave_sum3(double &a, double &b, double &c, bool odd, bool bit1) {
  auto dpp = [](double x, auto ctrl) -> double {
    constexpr int C = decltype(ctrl)::value;
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, C, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), C, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  };
  // lane ^ 1: even lanes keep (a, b) and send c; odd lanes keep c and send (a, b)
  const double k1 = odd ? c : a, k2 = odd ? 0.0 : b, s1 = odd ? a : c, s2 = odd ? b : 0.0;
  const double u1 = k1 + dpp(s1, std::integral_constant<int, 0xB1>{});      // even: a over the pair; odd: c over the pair
  const double u2 = k2 + dpp(s2, std::integral_constant<int, 0xB1>{});      // even: b over the pair; odd: 0
  // lane ^ 2: the lower pair of a quad keeps its first value and sends the second, the upper pair the other way round
  const double k = bit1 ? u2 : u1, s = bit1 ? u1 : u2;
  double v = k + dpp(s, std::integral_constant<int, 0x4E>{});               // quad lane 0: a, 1: c, 2: b, 3: 0 -- over the quad
  v += dpp(v, std::integral_constant<int, 0x124>{});                        // row_ror:4
  v += dpp(v, std::integral_constant<int, 0x128>{});                        // row_ror:8: over the 16-lane row, per quad lane
  {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)u, (unsigned)u, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);
    v = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]) + __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
  }
  {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)u, (unsigned)u, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);
    v = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]) + __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
  }
  // every lane must see the SAME bits (the skip / swap decisions are taken per lane): the lanes of a class added their row's four
  // quads in rotated orders, so the totals are read from lanes 0, 1, 2 rather than from each quad's own
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  auto from = [&](int l) -> double {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  };
  a = from(0);
  c = from(1);
  b = from(2);
}

// Sums of a, b, c over the 16 lanes of a DPP row, the SAME bits in every lane of the row (the lanes take a rotation's
// skip / swap decisions one by one): a butterfly whose two partners add the same two values -- x + y and y + x -- at every
// level (lane ^ 1, lane ^ 2, then the quads 0 <-> 1, 2 <-> 3 by row_half_mirror and the halves by row_mirror: every quad is
// uniform by then, so a mirror is a swap).  Four rows of a wavefront run four different rotations side by side.
__device__ __forceinline__ void row_sum3(double &a, double &b, double &c) {
  auto dpp = [](double x, auto ctrl) -> double {
    constexpr int C = decltype(ctrl)::value;
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    // (mov_dpp: no previous value to merge -- all rows and banks are written -- so no register has to be filled in first)
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u, C, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), C, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  };
  auto level = [&](auto ctrl) {
    a += dpp(a, ctrl);
    b += dpp(b, ctrl);
    c += dpp(c, ctrl);
  };
  level(std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  level(std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  level(std::integral_constant<int, 0x141>{});   // row_half_mirror
  level(std::integral_constant<int, 0x140>{});   // row_mirror
}

// the same over the 8 lanes of half a DPP row (three levels: lane ^ 1, lane ^ 2, the two quads by row_half_mirror)
__device__ __forceinline__ void octet_sum3(double &a, double &b, double &c) {
  auto dpp = [](double x, auto ctrl) -> double {
    constexpr int C = decltype(ctrl)::value;
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u, C, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), C, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  };
  auto level = [&](auto ctrl) {
    a += dpp(a, ctrl);
    b += dpp(b, ctrl);
    c += dpp(c, ctrl);
  };
  level(std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  level(std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  level(std::integral_constant<int, 0x141>{});   // row_half_mirror: the other quad of the octet
}

// ... and over the LPR = 8 or 4 lanes a rotation is laid over (4: the two levels inside a quad)
template <int LPR>
__device__ __forceinline__ void group_sum3(double &a, double &b, double &c) {
  if constexpr (LPR == 8) {
    octet_sum3(a, b, c);
  } else {
    static_assert(LPR == 4, "a rotation over 8 or 4 lanes");
    auto dpp = [](double x, auto ctrl) -> double {
      constexpr int C = decltype(ctrl)::value;
      const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
      const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u, C, 0xf, 0xf, false);
      const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), C, 0xf, 0xf, false);
      return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    };
    auto level = [&](auto ctrl) {
      a += dpp(a, ctrl);
      b += dpp(b, ctrl);
      c += dpp(c, ctrl);
    };
    level(std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
    level(std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  }
}

__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct HparmaParams {
  SpectroParams s;          // stream, frame0, nframes, H, R, history_mode, fmt, psd (taps/tw unused)
  int n;                    // block size N
  int t;                    // equations (rows)
  int ncol;                 // p_e + 1
  const int *sched;         // [nsteps][8]: the Jacobi sweep as steps of up to eight column-disjoint rotations, j | k << 8 or -1 (glfer_hip.cpp)
  int nsteps;
  int width;                // rotations per step the schedule was built for: 8 (a rotation over 8 lanes) or 16 (over 4)
  const uint16_t *lagmap;   // [t][ncol]: which autocorrelation lag each matrix cell ends up holding
  const float2 *unit;       // [N/2+1]: exp(-2 pi i k / N)
  unsigned *queue;          // frames handed out beyond the first gridDim.x (null: blockIdx.x, + gridDim.x, ...)
};

template <int FMT>
__device__ __forceinline__ float hp_sample(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  if constexpr (FMT == GLFER_FMT_F32) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 0, 0));
  else if constexpr (FMT == GLFER_FMT_S16) return (float)(short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, voff, 0, 0) / 32768.0f;
  else return ((float)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(rsrc, voff, 0, 0) - 128.0f) / 128.0f;
}

// LPR: lanes a scheduled rotation is laid over (8: eight rotations per step, the product; 4: sixteen -- a kernel of its own, so that
// its registers do not cost the other form its occupancy)
// TT, NC: the matrix shape as compile-time constants (BASELINE config 5: t = 128, p_e + 1 = 33), 0 = taken from the parameters.  With the
// shape known the step is straight-line code: no row tests around the loads and stores of a column, column offsets by shifts instead
// of v_mul_lo_u32, Q's five row slots instead of eight tested ones.
template <int FMT, int LPR = 8, int TT = 0, int NC = 0>
__global__ __launch_bounds__(64) void hparma_kernel(HparmaParams hp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SpectroParams &p = hp.s;
  const int N = hp.n, t = TT ? TT : hp.t, ncol = NC ? NC : hp.ncol;
  const int lane = threadIdx.x;
  typedef float v4f32 __attribute__((ext_vector_type(4)));
  constexpr unsigned esz = FMT == GLFER_FMT_F32 ? 4 : (FMT == GLFER_FMT_S16 ? 2 : 1);
  // LDS: [x (N floats + a zero tail of 128 when t <= 128) overlaid later by A (t*ncol floats)] [Q ncol*ncol] [r t] [S ncol] [a ncol]
  const int xlen = N + (t <= 128 ? 128 : 0);
  const int big = xlen > t * ncol ? xlen : t * ncol;
  float *x = smem, *A = smem, *Q = smem + big, *rl = Q + ncol * ncol, *S = rl + t, *ar = S + ncol;

  // Frames are handed out from a queue when a launch has more of them than wavefronts in flight (hp.queue, zeroed by the launcher):
  // seven one-wavefront workgroups share a CU's four SIMDs 2 + 2 + 2 + 1 (tools/ldsocc), the one alone on its SIMD is half as
  // fast again as the others, and with a fixed stride it sat idle for the last third of the launch.
  for (long long f = blockIdx.x; f < p.nframes;
       f = hp.queue ? (long long)gridDim.x + (long long)__builtin_amdgcn_readfirstlane(lane == 0 ? (int)atomicAdd(hp.queue, 1u) : 0)
                    : f + gridDim.x) {
    // ---- K1: the assembled frame (prepare_audio, fft.c:98-113), unwindowed (source.c:369)
    {
      const long long s0 = (p.frame0 + f) * (long long)p.H - p.R;
      const long long sbase = s0 > 0 ? s0 : 0;
      const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char *>(reinterpret_cast<const char *>(p.stream)) + sbase * (long long)esz, 0, 0x7fffffff, 0x00020000);
      const int rel0 = (int)(s0 - sbase);
      // sixteen loads in flight per lane and one wait for them (one load and its wait per iteration, as this loop used to be
      // compiled, is 64 trips to memory a frame -- with two wavefronts on a SIMD at best, nobody covers them)
      constexpr int UN = 16;
      for (int j0 = lane; j0 < N; j0 += 64 * UN) {
        float v[UN];
        bool ok[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
          const int j = j0 + 64 * u, rel = rel0 + j;
          ok[u] = j < N && (p.history_mode ? (j >= p.R) : (rel >= 0));
          v[u] = hp_sample<FMT>(xrsrc, ok[u] ? (unsigned)rel * esz : 0x80000000u);
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
          const int j = j0 + 64 * u;
          if (j < N) x[j] = ok[u] ? v[u] : 0.0f;
        }
      }
      if (t <= 128) {
        x[N + lane] = 0.0f;
        x[N + 64 + lane] = 0.0f;
      }
    }
    wave_fence();
    // ---- autocorrelation (hparma.c:89-95): float products summed in double, k ascending
    if (t <= 128) {
      // Lags `lane` and `lane + 64` side by side, four k at a time: x[k .. k+3] is one broadcast ds_read_b128 for both, the lane's own
      // operands x[k + i ..] two 4-byte-aligned ds_read2_b32 a lag, everything requested one round ahead.  Every lane walks all N
      // values of k: past its own N - i the frame's zero tail gives products of (+-)0, and s + (+-)0 is s bit for bit (s starts at
      // +0 and can never become -0), so the sum is hparma.c:91-93's.  The one-term-at-a-time loop this replaces was a quarter of the
      // kernel's issued instructions (14 a term, each iteration waiting for its own two LDS reads).
      const float *xa = x + lane, *xb = x + lane + 64;
      double s0 = 0.0, s1 = 0.0;
      v4f32 xk = *reinterpret_cast<const v4f32 *>(x);
      float a[4], b[4];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        a[c] = xa[c];
        b[c] = xb[c];
      }
      for (int k = 0; k < N; k += 4) {
        const int kn = k + 4 < N ? k + 4 : k;             // (the last round asks for its own values again)
        const v4f32 xkn = *reinterpret_cast<const v4f32 *>(x + kn);
        float an[4], bn[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
          an[c] = xa[kn + c];
          bn[c] = xb[kn + c];
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
          s0 += (double)(a[c] * xk[c]);
          s1 += (double)(b[c] * xk[c]);
        }
        xk = xkn;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          a[c] = an[c];
          b[c] = bn[c];
        }
      }
      if (lane < t) rl[lane] = (float)(s0 / (double)(N - lane));
      if (lane + 64 < t) rl[lane + 64] = (float)(s1 / (double)(N - lane - 64));
    } else {
      for (int i = lane; i < t; i += 64) {
        double s = 0.0;
        const int len = N - i;
        for (int k = 0; k < len; k++) s += (double)(x[k + i] * x[k]);
        rl[i] = (float)(s / (double)len);
      }
    }
    wave_fence();
    // ---- the t x ncol matrix, column-major A[j*t + i]; cell (i,j) holds lag lagmap[i][j]
    // (hparma.c:94,98-102 incl. the row-0 overflow), and Q = I (util.c:286-291)
    for (int i = lane; i < t; i += 64)
      for (int j = 0; j < ncol; j++) A[j * t + i] = rl[hp.lagmap[i * ncol + j]];
    for (int i = lane; i < ncol; i += 64)
      for (int j = 0; j < ncol; j++) Q[j * ncol + i] = (i == j) ? 1.0f : 0.0f;
    wave_fence();

    // ---- one-sided Jacobi (util.c:294-356), cyclic by columns, same order as the reference.
    // Every lane only ever touches its own rows (i = lane + 64 r) of A and its own row of Q, so LDS
    // is per-lane storage here (dynamic column index) and needs no fences inside a sweep.  Fast
    // path (t <= 128, p_e+1 <= 64: two rows of A and one of Q per lane): column j stays in VGPRs
    // for the whole k loop and column k+1 is fetched while column k's three sums are reduced.
    const int sweepmax = ncol > 12 ? ncol : 12;
    int count = 1, sweep = 0;
    const bool fast = t <= 128 && ncol <= 64;
    const bool odd = (lane & 1) != 0, bit1 = (lane & 2) != 0;
    // Round 4: COLUMN-DISJOINT ROTATIONS RUN SIDE BY SIDE.  compute_svd's rotation (j, k) touches columns j and k of A
    // and Q only, its skip tests are local to the pair and `count` is a per-sweep sum (util.c:301-355): the only order that
    // matters is the reference's row-cyclic order AMONG ROTATIONS THAT SHARE A COLUMN; rotations that share none commute
    // exactly.  (Anti-diagonals j + k = d are such sets: (j', j), (j, k') and (j', k) with j' < j, k' < k all have a
    // smaller index sum than (j, k), the later ones a larger.)  The host turns the walk into a static schedule of steps of
    // up to EIGHT column-disjoint rotations (glfer_hip.cpp: list scheduling of the dependency graph by longest remaining
    // path: 80 steps per sweep for 33 columns instead of 528 rotations one after the other; anti-diagonal by
    // anti-diagonal, four at a time -- the first form of this round -- it was 156).  A rotation is laid over the 8 lanes
    // of half a DPP row: a lane holds sixteen rows of the two columns (rows 32 c + 4 l .. + 3, c = 0..3: four
    // ds_read_b128 a column) and rows l, l + 8 ... of Q; the sums need three octet-local butterfly levels whose partners
    // add the same two values (x + y and y + x: the same bits in every lane, no read-back) instead of six wave-wide ones,
    // and the ~80 double-precision instructions of the angle (two divisions, two square roots) are issued once for eight
    // rotations.  Every rotation's own arithmetic is what it was; the sums associate differently (sixteen rows in a lane,
    // then the butterfly), as they already differed from the reference's row-by-row order.  t a multiple of 4, <= 128.
    const bool diag = hp.sched != nullptr && hp.nsteps > 0 && hp.width == 64 / LPR && t <= 128 && ncol <= 64 && (t & 3) == 0 && ncol >= 2;
    // one sweep of the schedule with a rotation over LPR lanes (64 / LPR rotations per step; hp.width = 64 / LPR is the width the host
    // scheduled for): a lane holds 128 / LPR rows of the two columns in chunks of four (rows (4 LPR) c + 4 l .. + 3) and rows l, l + LPR ... of Q
    auto sweep_scheduled = [&] {
      constexpr int NG = 64 / LPR, NCH = 32 / LPR, NQ = 64 / LPR;
      const int grp = lane / LPR, ll = lane % LPR;
      // Columns start 4 t bytes apart -- a multiple of the 256 bytes the 64 banks span for t = 64, 128 -- so the same chunk of two
      // columns lies on the same banks, and the two groups a ds_read_b128 serves together collided on every access (rocprofv3: 38 %
      // of the LDS array's active cycles were bank conflicts).  Odd groups therefore take their chunks in the order 1, 0, 3, 2: the
      // neighbours' 128 bytes fall on different halves of the banks.  (A lane's sixteen rows are the same; the order it adds
      // them in depends on its group's parity.)
      const int swz = LPR == 8 ? (grp & 1) : 0;
      int skipped = 0;
      int e = hp.sched[grp];
      for (int s0 = 0; s0 < hp.nsteps; s0++) {
        const int en = hp.sched[(s0 + 1 < hp.nsteps ? s0 + 1 : s0) * NG + grp];     // the next step's pair, requested a step ahead
        const bool act = e >= 0;
        const int j = act ? (e & 0xff) : 0, k = act ? (e >> 8) : 1;
        float *Aj = A + j * t, *Ak = A + k * t, *Qj = Q + j * ncol, *Qk = Q + k * ncol;
        const v4f32 z4 = v4f32{0.0f, 0.0f, 0.0f, 0.0f};
        v4f32 aj[NCH], ak[NCH];
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          const int r = 4 * LPR * (c ^ swz) + 4 * ll;
          aj[c] = r < t ? *reinterpret_cast<const v4f32 *>(Aj + r) : z4;
          ak[c] = r < t ? *reinterpret_cast<const v4f32 *>(Ak + r) : z4;
        }
        float qj[NQ], qk[NQ];
#pragma unroll
        for (int m = 0; m < NQ; m++) {                   // (unconditional loads of a clamped row: a branch per load would wait for each in turn)
          if (LPR * m < ncol) {                          // (uniform)
            const int qr = ll + LPR * m < ncol ? ll + LPR * m : ncol - 1;
            qj[m] = Qj[qr];
            qk[m] = Qk[qr];
          } else {
            qj[m] = qk[m] = 0.0f;
          }
        }
        double pp = 0.0, qq = 0.0, rr = 0.0;            // (float products are exact in double: fma and multiply + add round alike)
#pragma unroll
        for (int c = 0; c < NCH; c++) {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const double a = aj[c][i], b = ak[c][i];
            pp = __builtin_fma(a, b, pp);
            qq = __builtin_fma(a, a, qq);
            rr = __builtin_fma(b, b, rr);
          }
        }
        group_sum3<LPR>(pp, qq, rr);
        bool rotate = act;
        if (qq * rr < 2.22e-16) rotate = false;                                    // util.c:316-320
        else if (pp * pp < 1.0e-12 * (qq * rr)) rotate = false;                    // util.c:321-325, without the division (qq * rr > 0 here)
        skipped += (act && !rotate) ? 1 : 0;
        if (rotate) {
          double cs, sn;
          if (qq < rr) {                                                            // util.c:327-335
            cs = 0.0;
            sn = 1.0;
          } else {
            qq -= rr;
            const double v = sqrt(4.0 * pp * pp + qq * qq);
            cs = sqrt((v + qq) / (2.0 * v));
            sn = pp / (v * cs);
          }
          auto turn = [&](float xj, float xk, float &oj, float &ok) {              // util.c:338-350
            const double a = xj, b = xk;
            oj = (float)(a * cs + b * sn);
            ok = (float)(-a * sn + b * cs);
          };
#pragma unroll
          for (int c = 0; c < NCH; c++) {
            const int r = 4 * LPR * (c ^ swz) + 4 * ll;
            if (r < t) {
              float nj[4], nk[4];
#pragma unroll
              for (int i = 0; i < 4; i++) turn(aj[c][i], ak[c][i], nj[i], nk[i]);
              *reinterpret_cast<v4f32 *>(Aj + r) = v4f32{nj[0], nj[1], nj[2], nj[3]};
              *reinterpret_cast<v4f32 *>(Ak + r) = v4f32{nk[0], nk[1], nk[2], nk[3]};
            }
          }
#pragma unroll
          for (int m = 0; m < NQ; m++) {
            if (LPR * m < ncol) {                                                   // (uniform: slots past the last row of Q are skipped)
              float oj, ok;
              turn(qj[m], qk[m], oj, ok);
              if (ll + LPR * m < ncol) {
                Qj[ll + LPR * m] = oj;
                Qk[ll + LPR * m] = ok;
              }
            }
          }
        }
        wave_fence();                                    // the next step's rotations read these columns from other lanes
        e = en;
      }
      // `count` (util.c:298, 318, 323): the sweep's pairs minus the skipped ones, over the groups
      int sk = 0;
#pragma unroll
      for (int g = 0; g < NG; g++) sk += __builtin_amdgcn_readlane(skipped, LPR * g);
      count = ncol * (ncol - 1) / 2 - sk;
    };
    while (diag && count > 0 && sweep <= sweepmax) {
      sweep_scheduled();
      sweep++;
    }
    while (!diag && count > 0 && sweep <= sweepmax) {
      count = ncol * (ncol - 1) / 2;
      if (fast) {
        const int i0 = lane, i1 = lane + 64;
        const bool v0 = i0 < t, v1 = i1 < t, vq = lane < ncol;
        for (int j = 0; j < ncol - 1; j++) {
          float aj0 = v0 ? A[j * t + i0] : 0.0f, aj1 = v1 ? A[j * t + i1] : 0.0f;
          float qj = vq ? Q[j * ncol + lane] : 0.0f;
          float ak0 = v0 ? A[(j + 1) * t + i0] : 0.0f, ak1 = v1 ? A[(j + 1) * t + i1] : 0.0f;
          float qk = vq ? Q[(j + 1) * ncol + lane] : 0.0f;
          for (int k = j + 1; k < ncol; k++) {
            float an0 = 0.0f, an1 = 0.0f, qn = 0.0f;                   // column k+1, in flight under the sums
            if (k + 1 < ncol) {
              an0 = v0 ? A[(k + 1) * t + i0] : 0.0f;
              an1 = v1 ? A[(k + 1) * t + i1] : 0.0f;
              qn = vq ? Q[(k + 1) * ncol + lane] : 0.0f;
            }
            double pp, qq, rr;                                       // the lane's two rows (0 + x is x: no addition for the first)
            {
              const double a = aj0, b = ak0, a1 = aj1, b1 = ak1;
              pp = a * b + a1 * b1;
              qq = a * a + a1 * a1;
              rr = b * b + b1 * b1;
            }
            wave_sum3(pp, qq, rr, odd, bit1);
            bool rotate = true;
            if (qq * rr < 2.22e-16) { count--; rotate = false; }                       // util.c:316-320
            else if (pp * pp < 1.0e-12 * (qq * rr)) { count--; rotate = false; }       // util.c:321-325: p*p/(q*r) < 1e-12, without the division (qq * rr > 0 here)
            if (rotate) {
              double cs, sn;
              if (qq < rr) {                                          // util.c:327-335
                cs = 0.0;
                sn = 1.0;
              } else {
                qq -= rr;
                const double v = sqrt(4.0 * pp * pp + qq * qq);
                cs = sqrt((v + qq) / (2.0 * v));
                sn = pp / (v * cs);
              }
              {                                                       // util.c:338-343
                const double a = aj0, b = ak0;
                aj0 = (float)(a * cs + b * sn);
                ak0 = (float)(-a * sn + b * cs);
              }
              {
                const double a = aj1, b = ak1;
                aj1 = (float)(a * cs + b * sn);
                ak1 = (float)(-a * sn + b * cs);
              }
              {                                                       // util.c:345-350
                const double a = qj, b = qk;
                qj = (float)(a * cs + b * sn);
                qk = (float)(-a * sn + b * cs);
              }
              if (v0) A[k * t + i0] = ak0;
              if (v1) A[k * t + i1] = ak1;
              if (vq) Q[k * ncol + lane] = qk;
            }
            ak0 = an0;
            ak1 = an1;
            qk = qn;
          }
          if (v0) A[j * t + i0] = aj0;
          if (v1) A[j * t + i1] = aj1;
          if (vq) Q[j * ncol + lane] = qj;
        }
        wave_fence();
      } else {
      for (int j = 0; j < ncol - 1; j++) {
        for (int k = j + 1; k < ncol; k++) {
          double pp = 0.0, qq = 0.0, rr = 0.0;
          for (int i = lane; i < t; i += 64) {
            const double aj = A[j * t + i], ak = A[k * t + i];
            pp += aj * ak;
            qq += aj * aj;
            rr += ak * ak;
          }
          wave_sum3(pp, qq, rr, odd, bit1);
          if (qq * rr < 2.22e-16) { count--; continue; }            // util.c:316-320
          if (pp * pp < 1.0e-12 * (qq * rr)) { count--; continue; } // util.c:321-325 (see the fast path)
          double cs, sn;
          if (qq < rr) {                                            // util.c:327-335
            cs = 0.0;
            sn = 1.0;
          } else {
            qq -= rr;
            const double v = sqrt(4.0 * pp * pp + qq * qq);
            cs = sqrt((v + qq) / (2.0 * v));
            sn = pp / (v * cs);
          }
          for (int i = lane; i < t; i += 64) {                      // util.c:338-343
            const double ak = A[k * t + i], aj = A[j * t + i];
            A[j * t + i] = (float)(aj * cs + ak * sn);
            A[k * t + i] = (float)(-aj * sn + ak * cs);
          }
          for (int i = lane; i < ncol; i += 64) {                   // util.c:345-350
            const double qj = Q[j * ncol + i], qk = Q[k * ncol + i];
            Q[j * ncol + i] = (float)(qj * cs + qk * sn);
            Q[k * ncol + i] = (float)(-qj * sn + qk * cs);
          }
          wave_fence();
        }
      }
      }
      sweep++;
    }
    // ---- singular values (util.c:365-373)
    for (int j = 0; j < ncol; j++) {
      double q = 0.0;
      for (int i = lane; i < t; i += 64) {
        const double a = A[j * t + i];
        q += a * a;
      }
      q = wave_sum(q);
      if (lane == 0) S[j] = (float)sqrt(q);
    }
    wave_fence();
    // ---- rank (hparma.c:106-122) -- every lane repeats it, in the reference's order
    double sum_sigma2 = 0.0;
    for (int i = 0; i < ncol; i++) sum_sigma2 += (double)(S[i] * S[i]);
    int prank = 4;
    {
      double acc = 0.0;
      for (int i = 0; i < ncol; i++) {
        acc += (double)(S[i] * S[i]);
        if (sqrt(acc / sum_sigma2) > 0.995) { prank = i; break; }
      }
    }
    // ---- AR vector from the noise subspace (hparma.c:125-138); v[i][k] = Q[k*ncol + i]
    const int p_e = ncol - 1;
    for (int i = lane; i < ncol; i += 64) {
      double num = 0.0, den = 0.0;
      for (int k = prank + 1; k <= p_e; k++) {
        num += (double)(Q[k * ncol + 0] * Q[k * ncol + i]);
        den += (double)(Q[k * ncol + 0] * Q[k * ncol + 0]);
      }
      ar[i] = (prank < p_e) ? (float)(num / den) : (i == 0 ? 1.0f : 0.0f);
    }
    wave_fence();
    // ---- |A(f)|^2/N by Horner at z = exp(-2 pi i k/N) (what the zero-padded N-point FFT of
    // hparma.c:140-153 evaluates), reciprocal below Nyquist (hparma.c:154-156)
    float *o = p.psd + (size_t)f * (size_t)p.pitch;
    for (int k = lane; k <= N / 2; k += 64) {
      const float2 z = hp.unit[k];
      // double Horner: the reciprocal below magnifies evaluation error at the spectral peaks
      const double zx = z.x, zy = z.y;
      double re = ar[p_e], im = 0.0;
      for (int m = p_e - 1; m >= 0; m--) {
        const double nr = re * zx - im * zy + (double)ar[m];
        im = re * zy + im * zx;
        re = nr;
      }
      const float ps = (float)((re * re + im * im) / (double)N);
      o[k] = (k < N / 2) ? (float)(1.0 / (double)ps) : ps;
    }
    wave_fence();
  }
}

}  // namespace glfer

using namespace glfer;

extern "C" hipError_t glfer_launch_hparma(const SpectroParams *sp, int n, int t, int ncol, const int *rot_sched, int rot_steps, int rot_width,
                                          const uint16_t *lagmap, const float2 *unit, hipStream_t st) {
  if (sp->nframes <= 0) return hipSuccess;
  HparmaParams hp;
  hp.s = *sp;
  hp.n = n;
  hp.t = t;
  hp.ncol = ncol;
  hp.sched = rot_sched;
  hp.nsteps = rot_steps;
  hp.width = rot_width;
  hp.lagmap = lagmap;
  hp.unit = unit;
  const int xlen = n + (t <= 128 ? 128 : 0);            // the frame and its zero tail (the autocorrelation's two-lag walk)
  const int big = xlen > t * ncol ? xlen : t * ncol;
  size_t shmem = (size_t)(big + ncol * ncol + t + 2 * ncol) * sizeof(float);
  // GLFER_HPARMA_LDS_KB (tools/hparma_occupancy.sh only): ask for more LDS than the frame needs, i.e. fewer frames in flight per CU
  static const long lds_kb = [] { const char *e = getenv("GLFER_HPARMA_LDS_KB"); return e ? atol(e) : 0L; }();
  if (lds_kb > 0 && (size_t)lds_kb * 1024 > shmem && lds_kb <= 64) shmem = (size_t)lds_kb * 1024;
  const long long resident = 256LL * (shmem ? (160 * 1024) / shmem : 8);
  const unsigned grid = (unsigned)(sp->nframes < resident ? sp->nframes : resident);
  hipError_t e = hipSuccess;
  hp.queue = nullptr;
  if ((long long)sp->nframes > (long long)grid) {
    e = glfer::scratch_malloc((void **)&hp.queue, 256, st);
    if (e == hipSuccess) e = hipMemsetAsync(hp.queue, 0, sizeof(unsigned), st);
    if (e != hipSuccess) return e;
  }
#define GLFER_HPARMA_LAUNCH(F, L, TT, NC)                                                                                  \
  do {                                                                                                                      \
    e = glfer::allow_dynamic_lds((const void *)hparma_kernel<F, L, TT, NC>, shmem);                                         \
    if (e == hipSuccess) hipLaunchKernelGGL((hparma_kernel<F, L, TT, NC>), dim3(grid), dim3(64), shmem, st, hp);            \
  } while (0)
  const bool wide = rot_width == 16;
  // GLFER_HPARMA_GENERIC=1 (A/B runs and the tests): the shape from the parameters even for t = 128, p_e = 32
  const bool generic_only = [] { const char *e = getenv("GLFER_HPARMA_GENERIC"); return e && atoi(e) != 0; }();
  const bool fixed = !wide && !generic_only && rot_sched != nullptr && rot_steps > 0;
  const bool c5 = fixed && t == 128 && ncol == 33;                 // BASELINE config 5
  const bool dflt = fixed && t == 96 && ncol == 17;                // glfer's own defaults (glfer.c:248-249: t = 96, p_e = 16)
  switch (sp->fmt) {
#define GLFER_HPARMA_FMT(F)                                    \
    case F:                                                    \
      if (wide) GLFER_HPARMA_LAUNCH(F, 4, 0, 0);               \
      else if (c5) GLFER_HPARMA_LAUNCH(F, 8, 128, 33);         \
      else if (dflt) GLFER_HPARMA_LAUNCH(F, 8, 96, 17);        \
      else GLFER_HPARMA_LAUNCH(F, 8, 0, 0);                    \
      break;
    GLFER_HPARMA_FMT(GLFER_FMT_F32)
    GLFER_HPARMA_FMT(GLFER_FMT_S16)
    GLFER_HPARMA_FMT(GLFER_FMT_U8)
#undef GLFER_HPARMA_FMT
    default: e = hipErrorInvalidValue;
  }
  if (hp.queue) glfer::scratch_free(hp.queue, st);
#undef GLFER_HPARMA_LAUNCH
  if (e != hipSuccess) return e;
  return hipGetLastError();
}
