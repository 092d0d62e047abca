// GK(7,15) panel rule shared by the host driver (iai_host.cpp) and the device-side inner adaptive
// loop (kernels.hip): same table, same operation order, FMA contraction OFF, so that a panel gives
// bit-identical (I, E) on both sides and the heaps make identical decisions.
// ref: QuadGK.evalrule for order 7 as reached from src/algorithms.jl:215-239 (see oracle gk_evalrule).
#pragma once

#if defined(__HIPCC__)
#define ABZ_HD __host__ __device__ inline
#else
#define ABZ_HD inline
#endif

namespace abz {

struct gkc {  // plain complex pair (no operator overloads: every operation below is explicit)
    double re, im;
};

// Kronrod-15 abscissae on [-1, 0] (QuadGK ordering), weights, embedded Gauss-7 weights (QUADPACK qk15)
ABZ_HD double gk_x(int i) {
    constexpr double X[8] = {-0.991455371120812639206854697526329, -0.949107912342758524526189684047851,
                             -0.864864423359769072789712788640926, -0.741531185599394439863864773280788,
                             -0.586087235467691130294144838258730, -0.405845151377397166906606412076961,
                             -0.207784955007898467600689403773245, 0.0};
    return X[i];
}
ABZ_HD double gk_w(int i) {
    constexpr double W[8] = {0.022935322010529224963732008058970, 0.063092092629978553290700663189204,
                             0.104790010322250183839876322541518, 0.140653259715525918745189590510238,
                             0.169004726639267902826583426598550, 0.190350578064785409913256402421014,
                             0.204432940075298892414161999234649, 0.209482141084727828012999174891714};
    return W[i];
}
ABZ_HD double gk_gw(int i) {
    constexpr double GW[4] = {0.129484966168869693270611432679082, 0.279705391489276667901467771423780,
                              0.381830050505118944950369775488975, 0.417959183673469387755102040816327};
    return GW[i];
}

// node i (0..14) of the panel (a, b) in QuadGK's batch order: (a+(1+x_j)s, a+(1-x_j)s) for j=0..6, then a+s
ABZ_HD double gk15_node(double a, double b, int i) {
#pragma clang fp contract(off)
    const double s = 0.5 * (b - a);
    if (i == 14) return a + s;
    const double xj = gk_x(i >> 1);
    const double t = (i & 1) ? (1.0 - xj) : (1.0 + xj);
    const double ts = t * s;
    return a + ts;
}

// fv: 15 values x ncomp components (node-major, stride ncomp).  I[c] = I_K * s, returns E = ||I_K s - I_G s||_2
template <class CPTR, class OPTR>
ABZ_HD double gk15_rule(CPTR fv, int ncomp, double a, double b, OPTR I) {
#pragma clang fp contract(off)
    const double s = 0.5 * (b - a);
    double e2 = 0.0;
    for (int c = 0; c < ncomp; ++c) {
#define ABZ_F(i) fv[(i) * ncomp + c]
        // fg = F(2)+F(3); fk = F(0)+F(1); Ig = fg*gw0; Ik = fg*w1 + fk*w0
        double fgr = ABZ_F(2).re + ABZ_F(3).re, fgi = ABZ_F(2).im + ABZ_F(3).im;
        double fkr = ABZ_F(0).re + ABZ_F(1).re, fki = ABZ_F(0).im + ABZ_F(1).im;
        double Igr = fgr * gk_gw(0), Igi = fgi * gk_gw(0);
        double t1r = fgr * gk_w(1), t1i = fgi * gk_w(1);
        double t2r = fkr * gk_w(0), t2i = fki * gk_w(0);
        double Ikr = t1r + t2r, Iki = t1i + t2i;
        for (int i = 2; i < 4; ++i) {
            fgr = ABZ_F(2 * (2 * i - 1)).re + ABZ_F(2 * (2 * i - 1) + 1).re;
            fgi = ABZ_F(2 * (2 * i - 1)).im + ABZ_F(2 * (2 * i - 1) + 1).im;
            fkr = ABZ_F(2 * (2 * i - 2)).re + ABZ_F(2 * (2 * i - 2) + 1).re;
            fki = ABZ_F(2 * (2 * i - 2)).im + ABZ_F(2 * (2 * i - 2) + 1).im;
            const double gr = fgr * gk_gw(i - 1), gi = fgi * gk_gw(i - 1);
            Igr = Igr + gr;
            Igi = Igi + gi;
            const double ar = fgr * gk_w(2 * i - 1), ai = fgi * gk_w(2 * i - 1);
            const double br = fkr * gk_w(2 * i - 2), bi = fki * gk_w(2 * i - 2);
            Ikr = (Ikr + ar) + br;
            Iki = (Iki + ai) + bi;
        }
        const double f0r = ABZ_F(14).re, f0i = ABZ_F(14).im;
        {
            const double gr = f0r * gk_gw(3), gi = f0i * gk_gw(3);
            Igr = Igr + gr;
            Igi = Igi + gi;
            const double ar = f0r * gk_w(7), ai = f0i * gk_w(7);
            const double lr = ABZ_F(12).re + ABZ_F(13).re, li = ABZ_F(12).im + ABZ_F(13).im;
            const double br = lr * gk_w(6), bi = li * gk_w(6);
            Ikr = (Ikr + ar) + br;
            Iki = (Iki + ai) + bi;
        }
#undef ABZ_F
        const double Iksr = Ikr * s, Iksi = Iki * s;
        const double Igsr = Igr * s, Igsi = Igi * s;
        I[c].re = Iksr;
        I[c].im = Iksi;
        const double dr = Iksr - Igsr, di = Iksi - Igsi;
        const double d2r = dr * dr, d2i = di * di;
        const double nrm = d2r + d2i;  // std::norm
        e2 = e2 + nrm;
    }
    return sqrt(e2);
}

}  // namespace abz
