// Host-side validation of the fast binning estimate's acceptance margins (test tool, built by
// tests/test_binning_margins.py with g++).  For random points it compares the fast estimate,
// when it claims certainty, with the exact chain of csrc/nsc_math.h.
//   usage: binning_check N seed emin_deg emax_deg E elev_f64 [lean [generator]]
//   prints: n az_uncertain el_uncertain az_wrong el_wrong az_slack s_lo s_hi
//   lean = 1: check nsc_point_lean (the streaming loop of encode_fast_kernel) instead of nsc_point_pixel; an
//   "uncertain" point counts in az_uncertain, a certain one must carry the exact pixel.
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include "nsc_math.h"

static uint64_t s[2];
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t next() {
    uint64_t s0 = s[0], s1 = s[1], r = s0 + s1;
    s1 ^= s0; s[0] = rotl(s0, 24) ^ s1 ^ (s1 << 16); s[1] = rotl(s1, 37);
    return r;
}
static inline double u01() { return (next() >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char **argv)
{
    long n = argc > 1 ? atol(argv[1]) : 1000000;
    s[0] = 0x9E3779B97F4A7C15ull ^ (argc > 2 ? strtoull(argv[2], 0, 10) : 1); s[1] = 0xD1B54A32D192ED03ull;
    double emin = (argc > 3 ? atof(argv[3]) : -24.8) * M_PI / 180.0;
    double emax = (argc > 4 ? atof(argv[4]) : 2.0) * M_PI / 180.0;
    int E = argc > 5 ? atoi(argv[5]) : 16;
    int f64 = argc > 6 ? atoi(argv[6]) : 1;
    int lean = argc > 7 ? atoi(argv[7]) : 0;
    int only_mode = argc > 8 ? atoi(argv[8]) : -1;     // -1: mix of the four generators below
    NscBinParams bp = nsc_make_bin_params(E, emin, emax, 1.0f, 80.0f, f64);
    long azu = 0, elu = 0, azw = 0, elw = 0;
    for (long i = 0; i < n; ++i) {
        float x, y, z;
        int mode = (int)(next() % 4);
        if (only_mode >= 0) mode = only_mode;
        if (mode == 0) {                       // spherical like the bench clouds
            double az = (u01() * 2 - 1) * M_PI, el = (u01() * 100 - 50) * M_PI / 180, r = 0.5 + u01() * 89.5;
            x = (float)(r * cos(el) * cos(az)); y = (float)(r * cos(el) * sin(az)); z = (float)(r * sin(el));
        } else if (mode == 1) {                // uniform cube
            x = (float)((u01() * 2 - 1) * 80); y = (float)((u01() * 2 - 1) * 80); z = (float)((u01() * 2 - 1) * 40);
        } else if (mode == 2) {                // points sitting on/next to column and row edges
            int c = (int)(next() % 360), rr = (int)(next() % (E + 1));
            double az = -M_PI + c * (2 * M_PI / 360) + (u01() - 0.5) * 4e-6;
            double el = emin + rr * (emax - emin) / E + (u01() - 0.5) * 4e-6, r = 1 + u01() * 70;
            x = (float)(r * cos(el) * cos(az)); y = (float)(r * cos(el) * sin(az)); z = (float)(r * sin(el));
        } else {                               // tiny / huge ratios, axis-aligned
            double a = (u01() * 2 - 1) * 60, b = (u01() * 2 - 1) * 1e-4;
            if (next() & 1) { x = (float)a; y = (float)b; } else { x = (float)b; y = (float)a; }
            z = (float)((u01() * 2 - 1) * 30);
            if ((next() % 16) == 0) y = 0.0f;
            if ((next() % 16) == 0) x = -0.0f;
            if ((next() % 64) == 0) x = INFINITY;
            if ((next() % 64) == 0) y = NAN;
            if ((next() % 64) == 0) z = -INFINITY;
            if ((next() % 64) == 0) x = 3e25f;
        }
        // reference decision, exact chain with the clip and isfinite tests spelled out
        const float xs = nsc_clip_sq(x), ys = nsc_clip_sq(y), zs = nsc_clip_sq(z);
        const float sxy = xs + ys, ss = sxy + zs;
        const bool keep = isfinite(x) && isfinite(y) && isfinite(z) && ss >= bp.s_lo && ss <= bp.s_hi;
        int pix; float sv;
        if (lean) {
            if (!nsc_lean_ok(bp)) { printf("lean path not valid for these parameters\n"); return 2; }
            const int st = nsc_point_lean(x, y, z, bp, pix, sv);   // what encode_fast_kernel runs per point
            if ((st != 0) != keep) { ++azw; ++elw; continue; }
            if (!keep) continue;
            if (sv != ss) ++azw;
            const int epix = nsc_row_exact(z, sxy, bp) * NSC_A + nsc_col_exact(y, x);
            if (nsc_point_exact(x, y, z, bp) != epix) ++elw;       // the queue's resolver == the definition
            if (st == 2) { ++azu; continue; }
            if (pix % NSC_A != epix % NSC_A) ++azw;
            if (pix / NSC_A != epix / NSC_A) ++elw;
            continue;
        }
        const int fl = nsc_point_pixel(x, y, z, bp, pix, sv);      // what the kernel runs
        if ((fl != 0) != keep) { ++azw; ++elw; continue; }
        if (!keep) continue;
        const int ec = nsc_col_exact(y, x), er = nsc_row_exact(z, sxy, bp);
        if (fl & 2) ++azu;
        if (fl & 4) ++elu;
        if (pix % NSC_A != ec) ++azw;
        if (pix / NSC_A != er) ++elw;
        if (sv != ss) ++azw;
    }
    printf("%ld %ld %ld %ld %ld %.9g %.9g %.9g\n", n, azu, elu, azw, elw, (double)nsc_az_edge_slack(),
           (double)bp.s_lo, (double)bp.s_hi);
    return 0;
}
