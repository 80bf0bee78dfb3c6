// host_tables.h -- once-per-plan host tables (window, DPSS tapers, twiddles).
#pragma once
namespace glfer {
void make_window(int type, int n, float *w);                               // fft.c:309-360
bool make_dpss(int n, int kmax, double nw, double *tapers, double *sig);   // g-l_dpss.c:288-347
// mtm.c:76-83 (U0), mtm.c:124-136 (sum_U0_sqr and hn, float accumulators): tables of the F-test
void make_ftest_tables(int n, int kmax, const double *tapers, double *U0, float *hn, float *sum_U0_sqr);
void make_twiddles(int n, int lanes, float *tw_re_im);                     // [64][lanes] (cos,sin)
void make_palette(int palette, unsigned char colortab[768]);               // g_main.c:651-762
constexpr int kLogThrK = 400;                                               // |10 log10 x| <= 400: every normal float, with room
const double *log_thresholds();                                             // [2 K + 1]: where (short)(10.0*log10(x)) steps, g_main.c:1192-1196
int plan16_passes(int logn, int radix[4]);                                  // spectro16.hip schedule
int make_twiddles16(int logn, float *tw_re_im);                            // returns slots per lane; [slot][N/16] (cos,sin)
}  // namespace glfer
