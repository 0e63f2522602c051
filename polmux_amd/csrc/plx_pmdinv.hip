// plx_pmdinv.hip -- inverse PMD matrix of a link and its application to the field, on gfx950.
//
// Reference: /root/reference/inverse_pmd.m:91-141 (U accumulated trunk by trunk with update_U :143-161 and
// getmatR :164-168, overall GVD :130-134, Uinv = U^H :135-136, application in the frequency domain :139-145).
//
// MI355X design: the trunk-to-trunk basis changes matR (2x2, frequency independent) are formed once on the
// host; one lane per frequency then walks the trunks with the SU(2) recurrence of update_U, keeping only the
// first row (U11, U12) -- update_U itself forces U21 = -conj(U12), U22 = conj(U11) -- plus the scalar Hgvd, in the order
// the row pass of the four-step FFT visits the spectrum.  Applying Uinv is ONE filter pass of the SSFM FFT
// engine (three in-place sweeps) with the 2x2 product fused where exp(-i beta dz) normally sits.  Frames of a
// batch may carry their own waveplate draws (Monte-Carlo PMD).
#include "plx_internal.h"
#include "plx_fft.h"

#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

namespace {

struct Trunk {          // one update_U call
    double r11x, r11y, r12x, r12y;   // first row of the matrix passed as matR
    double db0;                      // birefringence at FN = 0 of this trunk
    int fiber;                       // which db1 column; < 0: l1 = l2 = 1 (reference change / last trunk)
    int pad_;
};

struct UArgs {
    cplx *u;                 // [F][N][3] (U11, U12, Hgvd), row-pass order
    const Trunk *trunks;     // [sets][ntr]
    const double *db1;       // [nfib][N], natural (fft) order
    const double *allgvd;    // [N] natural order, or null (options.gvd == 'no')
    int ntr, per_frame, p1, p2;
    size_t N;
};

__global__ __launch_bounds__(256) void k_pmd_u(UArgs a)
{
    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= a.N) return;
    const int f = blockIdx.y;
    const int N1 = 1 << a.p1, N2 = 1 << a.p2;
    const unsigned j = (unsigned)(pos / N2), i = (unsigned)(pos % N2);
    const size_t k = (size_t)plx_bitrev(j, a.p1) + (size_t)N1 * plx_bitrev(i, a.p2);   // frequency index of this slot
    const Trunk *tr = a.trunks + (a.per_frame ? (size_t)f * a.ntr : 0);
    cplx u11 = make_double2(1.0, 0.0), u12 = make_double2(0.0, 0.0);                    // U = I, :103
    for (int s = 0; s < a.ntr; s++) {
        const Trunk t = tr[s];
        cplx t11 = make_double2(t.r11x, t.r11y), t12 = make_double2(t.r12x, t.r12y);
        if (t.fiber >= 0) {
            const double deltabeta = 0.5 * (a.db1[(size_t)t.fiber * a.N + k] + t.db0);  // :111, :125
            double sn, cs;
            sincos(-deltabeta, &sn, &cs);                                               // l1 = fastexp(-deltabeta)
            const cplx l1 = make_double2(cs, sn);
            t11 = cmul(l1, t11);                                                        // matT(1,:) = l1*matR(1,:), :149
            t12 = cmul(l1, t12);
        }
        const cplx u21 = make_double2(-u12.x, u12.y), u22 = make_double2(u11.x, -u11.y); // :155-156
        const cplx n11 = cadd(cmul(t11, u11), cmul(t12, u21));                          // :153
        const cplx n12 = cadd(cmul(t11, u12), cmul(t12, u22));                          // :154
        u11 = n11; u12 = n12;
    }
    cplx h = make_double2(1.0, 0.0);
    if (a.allgvd) {                                                                     // Hgvd = fastexp(-allgvd), :133
        double sn, cs;
        sincos(-a.allgvd[k], &sn, &cs);
        h = make_double2(cs, sn);
    }
    cplx *o = a.u + 3 * ((size_t)f * a.N + pos);   // the scalar stays separate: Hgvd*U is unitary but no longer SU(2)
    o[0] = u11;
    o[1] = u12;
    o[2] = h;
}

typedef std::complex<double> zc;

// getmatR, inverse_pmd.m:164-168
void getmatR(double theta, double eps, zc R[2][2])
{
    const double c = cos(theta), s = sin(theta), ce = cos(eps), se = sin(eps);
    const zc th[2][2] = {{c, -s}, {s, c}};                                   // cos*sig0 - sin*sig3i
    const zc ep[2][2] = {{zc(ce, 0), zc(0, se)}, {zc(0, se), zc(ce, 0)}};    // complex(cos*sig0, sin*sig2)
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) R[i][j] = th[i][0] * ep[0][j] + th[i][1] * ep[1][j];
}

} // namespace

struct plx_pmdinv {
    int64_t N = 0;
    int max_frames = 0, sets = 0, ntr = 0, nfib = 0, apply_gvd = 1;
    plx_ssfm *fft = nullptr;
    cplx *d_u = nullptr;
    Trunk *d_trunks = nullptr;
    double *d_db1 = nullptr, *d_allgvd = nullptr;
    int u_valid_frames = 0;
};

static void pmdinv_free(plx_pmdinv *P)
{
    if (!P) return;
    if (P->fft) plx_ssfm_destroy(P->fft);
    if (P->d_u) (void)hipFree(P->d_u);
    if (P->d_trunks) (void)hipFree(P->d_trunks);
    if (P->d_db1) (void)hipFree(P->d_db1);
    if (P->d_allgvd) (void)hipFree(P->d_allgvd);
    delete P;
}

extern "C" int plx_pmdinv_create(plx_pmdinv **out, int64_t nfft, int max_frames)
{
    if (!out) PLX_FAIL(PLX_ERR_ARG, "plx_pmdinv_create: null argument");
    *out = nullptr;
    if (max_frames < 1) PLX_FAIL(PLX_ERR_ARG, "plx_pmdinv_create: max_frames must be >= 1");
    plx_pmdinv *P = new plx_pmdinv();
    P->N = nfft; P->max_frames = max_frames;
    std::vector<double> zeros((size_t)(nfft > 0 ? nfft : 1), 0.0);
    double gam0 = 0.0;
    plx_ssfm_desc sd;
    std::memset(&sd, 0, sizeof(sd));
    sd.nfft = nfft; sd.nfc = 1; sd.dual_pol = 1; sd.max_frames = max_frames;
    sd.dzmaxt = 1; sd.dphimaxt = 1; sd.length = 1; sd.nplates = 1; sd.gam = &gam0; sd.betat = zeros.data();
    sd.fls[1] = 1;   // a PMD-type plan: the matrix multiplier needs both polarisations of a bin in one row workgroup
    int rc = plx_ssfm_create(&P->fft, &sd);
    if (rc != PLX_OK) { pmdinv_free(P); return rc; }
    if (hipMalloc((void **)&P->d_u, sizeof(cplx) * 3 * (size_t)nfft * max_frames) != hipSuccess) {
        pmdinv_free(P);
        PLX_FAIL(PLX_ERR_HIP, "plx_pmdinv_create: device allocation failed");
    }
    *out = P;
    return PLX_OK;
}

extern "C" int plx_pmdinv_destroy(plx_pmdinv *P)
{
    pmdinv_free(P);
    return PLX_OK;
}

extern "C" int plx_pmdinv_set_link(plx_pmdinv *P, int nfibers, const int32_t *ntrunk, const double *db0,
                                   const double *theta, const double *epsilon, const double *lcorr,
                                   const double *betat, const double *db1, const double *mat, int apply_gvd,
                                   int nsets)
{
    if (!P || !ntrunk || !db0 || !theta || !epsilon || !lcorr || !betat || !db1)
        PLX_FAIL(PLX_ERR_ARG, "plx_pmdinv_set_link: null argument");
    if (nfibers < 1 || (nsets != 1 && nsets > P->max_frames) || nsets < 1)
        PLX_FAIL(PLX_ERR_ARG, "plx_pmdinv_set_link: nfibers >= 1 and nsets in {1 .. max_frames} required");
    int tot = 0;
    for (int n = 0; n < nfibers; n++) {
        if (ntrunk[n] < 1) PLX_FAIL(PLX_ERR_ARG, "plx_pmdinv_set_link: every fibre needs at least one trunk");
        tot += ntrunk[n];
    }
    const size_t N = (size_t)P->N;
    // update_U calls per set: [options.mat] + per fibre (first trunk, trunks 2..n, closing rotation)
    const int ntr = (mat ? 1 : 0) + tot + nfibers;
    std::vector<Trunk> tr((size_t)nsets * ntr);
    for (int s = 0; s < nsets; s++) {
        Trunk *t = tr.data() + (size_t)s * ntr;
        const double *sdb0 = db0 + (size_t)s * tot, *sth = theta + (size_t)s * tot, *sep = epsilon + (size_t)s * tot;
        int q = 0;
        auto put = [&](zc a, zc b, double d0, int fiber) {
            t[q].r11x = a.real(); t[q].r11y = a.imag(); t[q].r12x = b.real(); t[q].r12y = b.imag();
            t[q].db0 = d0; t[q].fiber = fiber; t[q].pad_ = 0; q++;
        };
        if (mat) put(zc(mat[0], mat[1]), zc(mat[2], mat[3]), 0.0, -1);        // :105-107 (mat: row-major re,im pairs)
        int off = 0;
        for (int n = 0; n < nfibers; n++) {
            zc R[2][2], R1[2][2], R2[2][2];
            getmatR(sth[off], sep[off], R);
            put(std::conj(R[0][0]), std::conj(R[1][0]), sdb0[off], n);       // matR' first row, :110-114
            for (int k = 1; k < ntrunk[n]; k++) {
                getmatR(sth[off + k - 1], sep[off + k - 1], R1);
                getmatR(sth[off + k], sep[off + k], R2);
                // matR = matR2'*matR1, first row: sum_m conj(R2[m][0]) * R1[m][j]           :116-118
                const zc a = std::conj(R2[0][0]) * R1[0][0] + std::conj(R2[1][0]) * R1[1][0];
                const zc b = std::conj(R2[0][0]) * R1[0][1] + std::conj(R2[1][0]) * R1[1][1];
                put(a, b, sdb0[off + k], n);                                  // :125-128
            }
            getmatR(sth[off + ntrunk[n] - 1], sep[off + ntrunk[n] - 1], R);
            put(R[0][0], R[0][1], 0.0, -1);                                   // last trunk, :130-131
            off += ntrunk[n];
        }
    }
    std::vector<double> allgvd(N, 0.0);
    for (int n = 0; n < nfibers; n++)
        for (size_t k = 0; k < N; k++) allgvd[k] = allgvd[k] + betat[(size_t)n * N + k] * lcorr[n] * ntrunk[n];   // :132
    if (P->d_trunks) { (void)hipFree(P->d_trunks); P->d_trunks = nullptr; }
    if (P->d_db1) { (void)hipFree(P->d_db1); P->d_db1 = nullptr; }
    if (P->d_allgvd) { (void)hipFree(P->d_allgvd); P->d_allgvd = nullptr; }
    bool ok = hipMalloc((void **)&P->d_trunks, tr.size() * sizeof(Trunk)) == hipSuccess &&
              hipMemcpy(P->d_trunks, tr.data(), tr.size() * sizeof(Trunk), hipMemcpyHostToDevice) == hipSuccess &&
              hipMalloc((void **)&P->d_db1, (size_t)nfibers * N * sizeof(double)) == hipSuccess &&
              hipMemcpy(P->d_db1, db1, (size_t)nfibers * N * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
              hipMalloc((void **)&P->d_allgvd, N * sizeof(double)) == hipSuccess &&
              hipMemcpy(P->d_allgvd, allgvd.data(), N * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) PLX_FAIL(PLX_ERR_HIP, "plx_pmdinv_set_link: device allocation/upload failed");
    P->sets = nsets; P->ntr = ntr; P->nfib = nfibers; P->apply_gvd = apply_gvd ? 1 : 0;
    P->u_valid_frames = 0;
    return PLX_OK;
}

static int pmdinv_build(plx_pmdinv *P, int nframes, hipStream_t st)
{
    if (!P->d_trunks) PLX_FAIL(PLX_ERR_ARG, "plx_pmdinv: set_link has not been called");
    if (P->sets != 1 && P->sets < nframes) PLX_FAIL(PLX_ERR_ARG, "plx_pmdinv: fewer waveplate sets than frames");
    UArgs a;
    a.u = P->d_u; a.trunks = P->d_trunks; a.db1 = P->d_db1; a.allgvd = P->apply_gvd ? P->d_allgvd : nullptr;
    a.ntr = P->ntr; a.per_frame = P->sets != 1; a.N = (size_t)P->N;
    plx_ssfm_geometry(P->fft, &a.p1, &a.p2);
    PLX_LAUNCH(k_pmd_u, dim3((unsigned)((P->N + 255) / 256), (unsigned)nframes), dim3(256), 0, st, a);
    PLX_HIP(hipGetLastError());
    P->u_valid_frames = nframes;
    return PLX_OK;
}

extern "C" int plx_pmdinv_apply_dev(plx_pmdinv *P, double *d_ux, double *d_uy, int nframes, void *stream)
{
    if (!P || !d_ux || !d_uy) PLX_FAIL(PLX_ERR_ARG, "plx_pmdinv_apply_dev: null argument");
    if (nframes < 1 || nframes > P->max_frames) PLX_FAIL(PLX_ERR_ARG, "plx_pmdinv_apply_dev: nframes outside [1, max_frames]");
    int rc = pmdinv_build(P, nframes, (hipStream_t)stream);
    if (rc != PLX_OK) return rc;
    return plx_ssfm_filter_dev(P->fft, (cplx *)d_ux, (cplx *)d_uy, nullptr, nframes, stream, P->d_u);   // :139-145
}

extern "C" int plx_pmdinv_matrices(plx_pmdinv *P, int frame, double *U, double *Uinv)
{
    if (!P || frame < 0 || frame >= P->max_frames) PLX_FAIL(PLX_ERR_ARG, "plx_pmdinv_matrices: bad argument");
    int rc = pmdinv_build(P, frame + 1, nullptr);
    if (rc != PLX_OK) return rc;
    const size_t N = (size_t)P->N;
    std::vector<cplx> h(3 * N);
    PLX_HIP(hipDeviceSynchronize());
    PLX_HIP(hipMemcpy(h.data(), P->d_u + 3 * (size_t)frame * N, 3 * N * sizeof(cplx), hipMemcpyDeviceToHost));
    int p1, p2;
    plx_ssfm_geometry(P->fft, &p1, &p2);
    const size_t N1 = (size_t)1 << p1, N2 = (size_t)1 << p2;
    // MATLAB 3-D layout [2][2][Nfft], column-major, interleaved complex: element (r,c,k) at ((k*2 + c)*2 + r)
    for (size_t pos = 0; pos < N; pos++) {
        const size_t k = (size_t)plx_bitrev((unsigned)(pos / N2), p1) + N1 * plx_bitrev((unsigned)(pos % N2), p2);
        const cplx u11 = h[3 * pos], u12 = h[3 * pos + 1], hg = h[3 * pos + 2];
        const cplx su[2][2] = {{u11, u12}, {make_double2(-u12.x, u12.y), make_double2(u11.x, -u11.y)}};     // :155-156
        double m[2][2][2];                                                                                  // U(r,c) = Hgvd * SU(2) part, :134
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 2; c++) {
                m[r][c][0] = hg.x * su[r][c].x - hg.y * su[r][c].y;
                m[r][c][1] = hg.x * su[r][c].y + hg.y * su[r][c].x;
            }
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 2; c++) {
                if (U) { U[2 * ((k * 2 + c) * 2 + r)] = m[r][c][0]; U[2 * ((k * 2 + c) * 2 + r) + 1] = m[r][c][1]; }
                if (Uinv) { Uinv[2 * ((k * 2 + c) * 2 + r)] = m[c][r][0]; Uinv[2 * ((k * 2 + c) * 2 + r) + 1] = -m[c][r][1]; }   // :135-136
            }
    }
    return PLX_OK;
}
