/* Plain-C client of the C ABI (include/abzhip.h): no Python, no C++ -- the way a host language binds it.
 * Config 1 of BASELINE.json: s(x) = cos(2 pi x) (coefficients [0.5, 0, 0.5], offset -2 .. here first = -1),
 * f = 1.3 s + 1 integrates to 1 per unit cell, through PTR (rule + reduce), the store-free sum, IAI, AutoPTR
 * and abz_eval_nodes.  Exit code 0 = all checks passed, 77 = no GPU (ABZ_ERR_NOGPU), anything else = failure. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "abzhip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != ABZ_OK) {                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, abz_last_error());     \
            return rc_ == ABZ_ERR_NOGPU ? 77 : 1;                                \
        }                                                                        \
    } while (0)

int main(void) {
    abz_ctx* ctx = NULL;
    CHECK(abz_ctx_create(0, &ctx));
    /* 1-D scalar series: coefficients (re, im) of e^{2 pi i m x}, m = -1, 0, 1 */
    const double coef[6] = {0.5, 0.0, 0.0, 0.0, 0.5, 0.0};
    const int32_t dims[1] = {3}, first[1] = {-1};
    const double period[1] = {1.0};
    abz_series* s = NULL;
    CHECK(abz_series_create(ctx, coef, 1, dims, first, period, 1, &s));

    /* H(k) at explicit nodes */
    const double k[3] = {0.0, 0.25, 0.5};
    double H[3 * 2];
    CHECK(abz_eval_nodes(s, k, 3, ABZ_WANT_H, H, NULL));
    if (fabs(H[0] - 1.0) > 1e-14 || fabs(H[2]) > 1e-14 || fabs(H[4] + 1.0) > 1e-14) {
        fprintf(stderr, "eval_nodes: %g %g %g\n", H[0], H[2], H[4]);
        return 2;
    }

    /* PTR: cached rule + reduction of f = 1.3 s + 1 */
    abz_rule* r = NULL;
    CHECK(abz_ptr_rule_build(s, 50, 0, NULL, NULL, ABZ_WANT_H, &r));
    const double p[2] = {1.3, 1.0};
    double out[2];
    CHECK(abz_rule_reduce(r, ABZ_F_LINEAR, p, 2, NULL, 0, 1, out));
    if (fabs(out[0] - 1.0) > 1e-14 || fabs(out[1]) > 1e-14) {
        fprintf(stderr, "rule_reduce: %.17g %.17g\n", out[0], out[1]);
        return 3;
    }
    CHECK(abz_rule_destroy(r));

    /* the same without materialising the rule (grids of more than 128 points) */
    CHECK(abz_ptr_sum(s, 400, 0, 400, ABZ_F_LINEAR, p, 2, NULL, 0, 1, out));
    if (fabs(out[0] - 1.0) > 1e-14) {
        fprintf(stderr, "ptr_sum: %.17g\n", out[0]);
        return 4;
    }

    /* IAI (nested Gauss-Kronrod, here one level) on [0, 1] */
    const double a[1] = {0.0}, b[1] = {1.0};
    double err = 0.0;
    int64_t nev = 0, npan = 0;
    CHECK(abz_iai_solve(s, ABZ_LIMS_CUBIC, a, b, ABZ_F_LINEAR, p, 2, 0.0, 1e-10, -1.0, 0, 0, out, &err, &nev, NULL, 0, &npan));
    if (fabs(out[0] - 1.0) > 1e-9 || nev < 15) {
        fprintf(stderr, "iai_solve: %.17g err %g numevals %lld\n", out[0], err, (long long)nev);
        return 5;
    }
    /* a sweep of 40 swept values through ONE handle: the library deals them to its lanes (views of the series on streams of
     * their own, ABZ_IAI_LANES); every result is the one of the solve run alone */
    {
        double sw[40], outs[80], errs[40], one[2], e1 = 0.0;
        int64_t nevs[40], n1 = 0;
        const double eta[1] = {0.3};
        for (int i = 0; i < 40; ++i) sw[i] = -0.9 + 0.045 * i;
        CHECK(abz_iai_solve_many(s, ABZ_LIMS_CUBIC, a, b, ABZ_F_DOS, eta, 1, sw, 40, 1e-8, -1.0, 0, 0, outs, errs, nevs, NULL, 0, NULL));
        const int probe[3] = {0, 17, 39};
        for (int k = 0; k < 3; ++k) {
            const int i = probe[k];
            CHECK(abz_iai_solve(s, ABZ_LIMS_CUBIC, a, b, ABZ_F_DOS, eta, 1, sw[i], 1e-8, -1.0, 0, 0, one, &e1, &n1, NULL, 0, NULL));
            if (one[0] != outs[2 * i] || one[1] != outs[2 * i + 1] || e1 != errs[i] || n1 != nevs[i]) {
                fprintf(stderr, "iai_solve_many[%d]: %.17g vs %.17g, numevals %lld vs %lld\n", i, outs[2 * i], one[0], (long long)nevs[i], (long long)n1);
                return 7;
            }
        }
    }
    /* AutoPTR: the whole p-adaptive loop in the library (grids 50, 100, ...): twice, the second solve from the kept rules */
    int64_t nev_auto = 0;
    int32_t npt_last = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(abz_autoptr_solve(s, NULL, 0, ABZ_F_LINEAR, p, 2, 0.0, 50, 50, 1e-10, -1.0, 0, 2, 1.0, out, &err, &nev_auto, &npt_last));
        if (fabs(out[0] - 1.0) > 1e-13 || nev_auto != 150 || npt_last != 100 || err > 1e-13) {
            fprintf(stderr, "autoptr_solve: %.17g err %g numevals %lld npt %d\n", out[0], err, (long long)nev_auto, (int)npt_last);
            return 6;
        }
    }
    CHECK(abz_series_drop_rules(s));
    CHECK(abz_series_destroy(s));
    CHECK(abz_ctx_destroy(ctx));
    printf("abi_smoke ok: numevals(IAI) = %lld, numevals(AutoPTR) = %lld, library version %d\n", (long long)nev, (long long)nev_auto, abz_version());
    return 0;
}
