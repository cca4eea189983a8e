/*
 * kwave_oracle.c — fp32 CPU restatement of the reference's per-time-step algorithm.
 *
 * TEST INFRASTRUCTURE ONLY (see kwave_oracle.h).  Never linked into or called by the product path.
 * Parity status: "parity unpinned" by the reference (it ships no tests/fixtures and cannot be built
 * here); pinned by the closed-form K1 test and the fp64 NumPy restatement (tests/test_oracle_*.py).
 *
 * The FFT is an in-repo mixed-radix Cooley-Tukey (no FFTW/MKL in the image); it implements the
 * contract of cuFFT's R2C/C2R (unnormalised, forward sign -i, Nx/2+1 bins along the fastest axis)
 * used at /root/reference/MatrixClasses/CufftComplexMatrix.cpp:82-130,508-534.
 *
 * Build: gcc -O3 -std=c11 -fopenmp -ffp-contract=off -fPIC -shared (oracle/Makefile).
 */
#define _GNU_SOURCE
#include "kwave_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_PI_2
#define M_PI_2 1.57079632679489661923
#endif
#ifndef M_LOG10E
#define M_LOG10E 0.43429448190325182765
#endif

typedef struct { float re, im; } cpx;

static inline cpx cmul(cpx a, cpx b)
{ /* cuCmulf: Utils/CudaUtils.cuh:159-163 semantics (no fma contraction: -ffp-contract=off) */
  cpx r;
  r.re = a.re * b.re - a.im * b.im;
  r.im = a.re * b.im + a.im * b.re;
  return r;
}
static inline cpx cscale(cpx a, float s) { cpx r = { a.re * s, a.im * s }; return r; }
static inline cpx cadd(cpx a, cpx b) { cpx r = { a.re + b.re, a.im + b.im }; return r; }
static inline cpx csub(cpx a, cpx b) { cpx r = { a.re - b.re, a.im - b.im }; return r; }

/* ------------------------------------------------------------------------------------------------
 * 1-D complex FFT plan: decimation-in-time, radices 4,2,3,5 + generic odd prime.
 * ---------------------------------------------------------------------------------------------- */
#define MAXFACTORS 32
typedef struct
{
  int  n;
  int  inverse;
  int  factors[2 * MAXFACTORS];
  cpx* tw; /* tw[k] = exp(-/+ 2 pi i k / n) */
} fft_plan;

static fft_plan* plan_create(int n, int inverse)
{
  fft_plan* p = (fft_plan*)malloc(sizeof(fft_plan));
  p->n       = n;
  p->inverse = inverse;
  p->tw      = (cpx*)malloc(sizeof(cpx) * (size_t)n);
  for (int k = 0; k < n; k++)
  {
    double ph = -2.0 * M_PI * (double)k / (double)n;
    if (inverse) ph = -ph;
    p->tw[k].re = (float)cos(ph);
    p->tw[k].im = (float)sin(ph);
  }
  /* factorise: 4s first, then 2,3,5,7,... */
  int  m = n, f = 4, i = 0;
  double floor_sqrt = floor(sqrt((double)n));
  do
  {
    while (m % f)
    {
      switch (f)
      {
        case 4: f = 2; break;
        case 2: f = 3; break;
        default: f += 2; break;
      }
      if (f > floor_sqrt) f = m;
    }
    m /= f;
    p->factors[i++] = f;
    p->factors[i++] = m;
  } while (m > 1);
  return p;
}
static void plan_destroy(fft_plan* p)
{
  if (!p) return;
  free(p->tw);
  free(p);
}

static void bfly2(cpx* out, size_t fstride, const fft_plan* st, int m)
{
  cpx*       o2 = out + m;
  const cpx* tw = st->tw;
  for (int k = 0; k < m; k++)
  {
    cpx t  = cmul(o2[k], tw[(size_t)k * fstride]);
    o2[k]  = csub(out[k], t);
    out[k] = cadd(out[k], t);
  }
}
static void bfly4(cpx* out, size_t fstride, const fft_plan* st, int m)
{
  const cpx* tw = st->tw;
  const int  m2 = 2 * m, m3 = 3 * m;
  for (int k = 0; k < m; k++)
  {
    cpx s0 = cmul(out[k + m], tw[(size_t)k * fstride]);
    cpx s1 = cmul(out[k + m2], tw[(size_t)k * fstride * 2]);
    cpx s2 = cmul(out[k + m3], tw[(size_t)k * fstride * 3]);
    cpx s5 = csub(out[k], s1);
    cpx a0 = cadd(out[k], s1);
    cpx s3 = cadd(s0, s2);
    cpx s4 = csub(s0, s2);
    out[k + m2] = csub(a0, s3);
    out[k]      = cadd(a0, s3);
    if (st->inverse)
    {
      out[k + m].re  = s5.re - s4.im;
      out[k + m].im  = s5.im + s4.re;
      out[k + m3].re = s5.re + s4.im;
      out[k + m3].im = s5.im - s4.re;
    }
    else
    {
      out[k + m].re  = s5.re + s4.im;
      out[k + m].im  = s5.im - s4.re;
      out[k + m3].re = s5.re - s4.im;
      out[k + m3].im = s5.im + s4.re;
    }
  }
}
static void bfly_generic(cpx* out, size_t fstride, const fft_plan* st, int m, int p)
{
  const cpx* tw = st->tw;
  const int  n  = st->n;
  cpx scratch[64];
  cpx* sc = (p <= 64) ? scratch : (cpx*)malloc(sizeof(cpx) * (size_t)p);
  for (int u = 0; u < m; u++)
  {
    int k = u;
    for (int q1 = 0; q1 < p; q1++) { sc[q1] = out[k]; k += m; }
    k = u;
    for (int q1 = 0; q1 < p; q1++)
    {
      size_t twidx = 0;
      cpx acc = sc[0];
      for (int q = 1; q < p; q++)
      {
        twidx += fstride * (size_t)k;
        if (twidx >= (size_t)n) twidx %= (size_t)n;
        acc = cadd(acc, cmul(sc[q], tw[twidx]));
      }
      out[k] = acc;
      k += m;
    }
  }
  if (sc != scratch) free(sc);
}

static void fft_work(cpx* out, const cpx* f, size_t fstride, size_t in_stride, const int* factors, const fft_plan* st)
{
  cpx*      out_beg = out;
  const int p = *factors++;
  const int m = *factors++;
  cpx*      out_end = out + (size_t)p * m;
  if (m == 1)
  {
    do { *out = *f; f += fstride * in_stride; } while (++out != out_end);
  }
  else
  {
    do
    {
      fft_work(out, f, fstride * p, in_stride, factors, st);
      f += fstride * in_stride;
    } while ((out += m) != out_end);
  }
  out = out_beg;
  switch (p)
  {
    case 2: bfly2(out, fstride, st, m); break;
    case 4: bfly4(out, fstride, st, m); break;
    default: bfly_generic(out, fstride, st, m, p); break;
  }
}
/* out-of-place: in (stride in_stride) -> out contiguous */
static inline void fft_exec(const fft_plan* st, const cpx* in, size_t in_stride, cpx* out)
{
  if (st->n == 1) { out[0] = in[0]; return; }
  fft_work(out, in, 1, in_stride, st->factors, st);
}

/* ------------------------------------------------------------------------------------------------
 * 3-D R2C / C2R built from 1-D passes.
 * ---------------------------------------------------------------------------------------------- */
#define COLBLK 8

/* complex transform along a strided axis, in place: `count` lines; line l starts at base(l) */
static void pass_strided(cpx* data, const fft_plan* pl, size_t n, size_t stride, size_t nlines_fast, size_t nslow,
                         size_t slow_stride)
{
  /* lines are indexed (s, f): start = s*slow_stride + f, f in [0,nlines_fast) contiguous */
#pragma omp parallel
  {
    cpx* in  = (cpx*)malloc(sizeof(cpx) * n * COLBLK);
    cpx* out = (cpx*)malloc(sizeof(cpx) * n);
#pragma omp for collapse(2) schedule(static)
    for (size_t s = 0; s < nslow; s++)
      for (size_t fb = 0; fb < (nlines_fast + COLBLK - 1) / COLBLK; fb++)
      {
        size_t f0 = fb * COLBLK;
        size_t nb = (f0 + COLBLK <= nlines_fast) ? COLBLK : nlines_fast - f0;
        cpx*   base = data + s * slow_stride + f0;
        for (size_t k = 0; k < n; k++)
          for (size_t b = 0; b < nb; b++) in[b * n + k] = base[k * stride + b];
        for (size_t b = 0; b < nb; b++)
        {
          fft_exec(pl, in + b * n, 1, out);
          for (size_t k = 0; k < n; k++) base[k * stride + b] = out[k];
        }
      }
    free(in);
    free(out);
  }
}

/* real rows -> half-spectrum rows; two real rows per complex FFT */
static void pass_x_r2c(const float* in, cpx* out, size_t nx, size_t nrows)
{
  const size_t nxc = nx / 2 + 1;
  fft_plan*    pl  = plan_create((int)nx, 0);
#pragma omp parallel
  {
    cpx* z = (cpx*)malloc(sizeof(cpx) * nx);
    cpx* Z = (cpx*)malloc(sizeof(cpx) * nx);
#pragma omp for schedule(static)
    for (size_t r = 0; r < (nrows + 1) / 2; r++)
    {
      const size_t ra = 2 * r, rb = 2 * r + 1;
      const int    two = rb < nrows;
      const float* a = in + ra * nx;
      const float* b = two ? in + rb * nx : NULL;
      for (size_t k = 0; k < nx; k++) { z[k].re = a[k]; z[k].im = two ? b[k] : 0.0f; }
      fft_exec(pl, z, 1, Z);
      cpx* oa = out + ra * nxc;
      cpx* ob = two ? out + rb * nxc : NULL;
      for (size_t k = 0; k < nxc; k++)
      {
        cpx zk = Z[k];
        cpx zn = Z[(nx - k) % nx];
        oa[k].re = 0.5f * (zk.re + zn.re);
        oa[k].im = 0.5f * (zk.im - zn.im);
        if (two)
        {
          ob[k].re = 0.5f * (zk.im + zn.im);
          ob[k].im = 0.5f * (zn.re - zk.re);
        }
      }
    }
    free(z);
    free(Z);
  }
  plan_destroy(pl);
}

/* half-spectrum rows -> real rows (unnormalised inverse); two rows per complex FFT */
static void pass_x_c2r(const cpx* in, float* out, size_t nx, size_t nrows)
{
  const size_t nxc = nx / 2 + 1;
  fft_plan*    pl  = plan_create((int)nx, 1);
#pragma omp parallel
  {
    cpx* z = (cpx*)malloc(sizeof(cpx) * nx);
    cpx* Z = (cpx*)malloc(sizeof(cpx) * nx);
#pragma omp for schedule(static)
    for (size_t r = 0; r < (nrows + 1) / 2; r++)
    {
      const size_t ra = 2 * r, rb = 2 * r + 1;
      const int    two = rb < nrows;
      const cpx*   a = in + ra * nxc;
      const cpx*   b = two ? in + rb * nxc : NULL;
      /* Z = A + i*B with A,B hermitian-extended; C2R ignores imag of DC / Nyquist like cuFFT does */
      for (size_t k = 0; k < nxc; k++)
      {
        cpx ak = a[k];
        cpx bk = two ? b[k] : (cpx){ 0.0f, 0.0f };
        if (k == 0 || (2 * k == nx)) { ak.im = 0.0f; bk.im = 0.0f; }
        z[k].re = ak.re - bk.im;
        z[k].im = ak.im + bk.re;
        if (k != 0 && 2 * k != nx)
        {
          z[nx - k].re = ak.re + bk.im;
          z[nx - k].im = -ak.im + bk.re;
        }
      }
      fft_exec(pl, z, 1, Z);
      float* oa = out + ra * nx;
      float* ob = two ? out + rb * nx : NULL;
      for (size_t k = 0; k < nx; k++)
      {
        oa[k] = Z[k].re;
        if (two) ob[k] = Z[k].im;
      }
    }
    free(z);
    free(Z);
  }
  plan_destroy(pl);
}

void kwo_fft_r2c_3d(const float* in, float* out_, uint64_t nx, uint64_t ny, uint64_t nz)
{
  cpx*         out = (cpx*)out_;
  const size_t nxc = nx / 2 + 1;
  pass_x_r2c(in, out, nx, ny * nz);
  if (ny > 1)
  {
    fft_plan* py = plan_create((int)ny, 0);
    pass_strided(out, py, ny, nxc, nxc, nz, nxc * ny);
    plan_destroy(py);
  }
  if (nz > 1)
  {
    fft_plan* pz = plan_create((int)nz, 0);
    pass_strided(out, pz, nz, nxc * ny, nxc * ny, 1, 0);
    plan_destroy(pz);
  }
}

void kwo_fft_c2r_3d(const float* in_, float* out, uint64_t nx, uint64_t ny, uint64_t nz)
{
  const size_t nxc = nx / 2 + 1;
  const size_t nc  = nxc * ny * nz;
  cpx*         tmp = (cpx*)malloc(sizeof(cpx) * nc);
  memcpy(tmp, in_, sizeof(cpx) * nc);
  if (nz > 1)
  {
    fft_plan* pz = plan_create((int)nz, 1);
    pass_strided(tmp, pz, nz, nxc * ny, nxc * ny, 1, 0);
    plan_destroy(pz);
  }
  if (ny > 1)
  {
    fft_plan* py = plan_create((int)ny, 1);
    pass_strided(tmp, py, ny, nxc, nxc, nz, nxc * ny);
    plan_destroy(py);
  }
  pass_x_c2r(tmp, out, nx, ny * nz);
  free(tmp);
}

/* 1-D real transforms along one axis of a 3-D array; output has n_axis/2+1 bins along that axis,
 * other axes unchanged, layout stays [z][y][x] order with the reduced axis shortened. */
void kwo_fft_r2c_1d(const float* in, float* out_, uint64_t nx, uint64_t ny, uint64_t nz, int axis)
{
  cpx* out = (cpx*)out_;
  if (axis == 0) { pass_x_r2c(in, out, nx, ny * nz); return; }
  const size_t n      = (axis == 1) ? ny : nz;
  const size_t nr     = n / 2 + 1;
  const size_t stride = (axis == 1) ? nx : nx * ny;
  fft_plan*    pl     = plan_create((int)n, 0);
  const size_t oy = (axis == 1) ? nr : ny;
  const size_t nslow = (axis == 1) ? nz : 1;
  const size_t nfast = (axis == 1) ? nx : nx * ny;
#pragma omp parallel
  {
    cpx* z = (cpx*)malloc(sizeof(cpx) * n);
    cpx* Z = (cpx*)malloc(sizeof(cpx) * n);
#pragma omp for collapse(2) schedule(static)
    for (size_t s = 0; s < nslow; s++)
      for (size_t f = 0; f < nfast; f++)
      {
        const float* src = in + s * (nx * ny) + f;
        for (size_t k = 0; k < n; k++) { z[k].re = src[k * stride]; z[k].im = 0.0f; }
        fft_exec(pl, z, 1, Z);
        cpx* dst = out + s * (nx * oy) + f;
        for (size_t k = 0; k < nr; k++) dst[k * stride] = Z[k];
      }
    free(z);
    free(Z);
  }
  plan_destroy(pl);
}

void kwo_fft_c2r_1d(const float* in_, float* out, uint64_t nx, uint64_t ny, uint64_t nz, int axis)
{
  const cpx* in = (const cpx*)in_;
  if (axis == 0) { pass_x_c2r(in, out, nx, ny * nz); return; }
  const size_t n      = (axis == 1) ? ny : nz;
  const size_t nr     = n / 2 + 1;
  const size_t stride = (axis == 1) ? nx : nx * ny;
  fft_plan*    pl     = plan_create((int)n, 1);
  const size_t iy = (axis == 1) ? nr : ny;
  const size_t nslow = (axis == 1) ? nz : 1;
  const size_t nfast = (axis == 1) ? nx : nx * ny;
#pragma omp parallel
  {
    cpx* z = (cpx*)malloc(sizeof(cpx) * n);
    cpx* Z = (cpx*)malloc(sizeof(cpx) * n);
#pragma omp for collapse(2) schedule(static)
    for (size_t s = 0; s < nslow; s++)
      for (size_t f = 0; f < nfast; f++)
      {
        const cpx* src = in + s * (nx * iy) + f;
        for (size_t k = 0; k < nr; k++)
        {
          cpx v = src[k * stride];
          if (k == 0 || 2 * k == n) v.im = 0.0f;
          z[k] = v;
          if (k != 0 && 2 * k != n) { z[n - k].re = v.re; z[n - k].im = -v.im; }
        }
        fft_exec(pl, z, 1, Z);
        float* dst = out + s * (nx * ny) + f;
        for (size_t k = 0; k < n; k++) dst[k * stride] = Z[k].re;
      }
    free(z);
    free(Z);
  }
  plan_destroy(pl);
}

/* ------------------------------------------------------------------------------------------------
 * Simulation state
 * ---------------------------------------------------------------------------------------------- */
struct kwo_sim
{
  kwo_problem pr;
  size_t nx, ny, nz, nxc, n, nc;
  uint64_t t;
  float fft_divider;
  /* state */
  float *p, *ux, *uy, *uz, *rhox, *rhoy, *rhoz, *duxdx, *duydy, *duzdz, *t1, *t2, *t3;
  cpx *cx, *cy, *cz;
  /* derived medium */
  float *c2, *dtrho0sgx, *dtrho0sgy, *dtrho0sgz, *tau, *eta; /* arrays or NULL */
  float c2_s, dtrho0sgx_s, dtrho0sgy_s, dtrho0sgz_s, tau_s, eta_s, dtrho0_s;
  float *kappa, *nabla1, *nabla2, *source_kappa;
};

static float* falloc(size_t n) { return (float*)calloc(n, sizeof(float)); }

/* KSpaceFirstOrderSolver.cpp:2404-2452 (generateKappa), :2460-2506 (generateSourceKappa) */
static void generate_kappa(kwo_sim* s, int source_variant)
{
  const kwo_problem* pr = &s->pr;
  const float dx2Rec = 1.0f / (pr->dx * pr->dx);
  const float dy2Rec = 1.0f / (pr->dy * pr->dy);
  const float dz2Rec = (s->nz > 1) ? 1.0f / (pr->dz * pr->dz) : 0.0f; /* :2409, :2464: 0 for a 2-D grid */
  const float cRefDtPi = pr->c_ref * pr->dt * (float)M_PI;
  const float nxRec = 1.0f / (float)s->nx;
  const float nyRec = 1.0f / (float)s->ny;
  const float nzRec = 1.0f / (float)s->nz;
  float* dst = source_variant ? s->source_kappa : s->kappa;
#pragma omp parallel for schedule(static)
  for (size_t z = 0; z < s->nz; z++)
  {
    const float zf = (float)z;
    float zPart = 0.5f - fabsf(0.5f - zf * nzRec);
    zPart = (zPart * zPart) * dz2Rec;
    for (size_t y = 0; y < s->ny; y++)
    {
      const float yf = (float)y;
      float yPart = 0.5f - fabsf(0.5f - yf * nyRec);
      yPart = (yPart * yPart) * dy2Rec;
      const float yzPart = zPart + yPart;
      for (size_t x = 0; x < s->nxc; x++)
      {
        const float xf = (float)x;
        float xPart = 0.5f - fabsf(0.5f - xf * nxRec);
        xPart = (xPart * xPart) * dx2Rec;
        float k = cRefDtPi * sqrtf(xPart + yzPart);
        const size_t i = (z * s->ny + y) * s->nxc + x;
        if (source_variant) dst[i] = cosf(k);
        else dst[i] = (k == 0.0f) ? 1.0f : sinf(k) / k;
      }
    }
  }
}

/* KSpaceFirstOrderSolver.cpp:2514-2577 (generateKappaAndNablas) */
static void generate_kappa_and_nablas(kwo_sim* s)
{
  const kwo_problem* pr = &s->pr;
  const float dxSqRec = 1.0f / (pr->dx * pr->dx);
  const float dySqRec = 1.0f / (pr->dy * pr->dy);
  const float dzSqRec = (s->nz > 1) ? 1.0f / (pr->dz * pr->dz) : 0.0f; /* :2518 */
  const float cRefDt2 = pr->c_ref * pr->dt * 0.5f;
  const float pi2 = (float)M_PI * 2.0f;
  const float nxRec = 1.0f / (float)s->nx;
  const float nyRec = 1.0f / (float)s->ny;
  const float nzRec = 1.0f / (float)s->nz;
  const float alphaPower = pr->alpha_power;
#pragma omp parallel for schedule(static)
  for (size_t z = 0; z < s->nz; z++)
  {
    const float zf = (float)z;
    float zPart = 0.5f - fabsf(0.5f - zf * nzRec);
    zPart = (zPart * zPart) * dzSqRec;
    for (size_t y = 0; y < s->ny; y++)
    {
      const float yf = (float)y;
      float yPart = 0.5f - fabsf(0.5f - yf * nyRec);
      yPart = (yPart * yPart) * dySqRec;
      const float yzPart = zPart + yPart;
      for (size_t x = 0; x < s->nxc; x++)
      {
        const float xf = (float)x;
        float xPart = 0.5f - fabsf(0.5f - xf * nxRec);
        xPart = (xPart * xPart) * dxSqRec;
        float k = pi2 * sqrtf(xPart + yzPart);
        float cRefK = cRefDt2 * k;
        const size_t i = (z * s->ny + y) * s->nxc + x;
        s->kappa[i] = (cRefK == 0.0f) ? 1.0f : sinf(cRefK) / cRefK;
        float n1 = powf(k, alphaPower - 2.0f);
        float n2 = powf(k, alphaPower - 1.0f);
        if (n1 == INFINITY) n1 = 0.0f;
        if (n2 == INFINITY) n2 = 0.0f;
        s->nabla1[i] = n1;
        s->nabla2[i] = n2;
      }
    }
  }
}

/* KSpaceFirstOrderSolver.cpp:2584-2643 (generateTauAndEta); uses c0 before squaring (:2612-2613) */
static void generate_tau_eta(kwo_sim* s)
{
  const kwo_problem* pr = &s->pr;
  const float alphaPower = pr->alpha_power;
  const float tanPi2AlphaPower = tanf((float)M_PI_2 * alphaPower);
  const float alphaNeperCoeff =
    (100.0f * powf(1.0e-6f / (2.0f * (float)M_PI), alphaPower)) / (20.0f * (float)M_LOG10E);
  if (!pr->alpha_coeff && !pr->c0)
  {
    const float alphaCoeff2 = 2.0f * pr->alpha_coeff_s * alphaNeperCoeff;
    s->tau_s = (-alphaCoeff2) * powf(pr->c0_s, alphaPower - 1);
    s->eta_s = alphaCoeff2 * powf(pr->c0_s, alphaPower) * tanPi2AlphaPower;
    return;
  }
  s->tau = falloc(s->n);
  s->eta = falloc(s->n);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < s->n; i++)
  {
    const float alphaCoeff2 = 2.0f * alphaNeperCoeff * (pr->alpha_coeff ? pr->alpha_coeff[i] : pr->alpha_coeff_s);
    const float c0 = pr->c0 ? pr->c0[i] : pr->c0_s;
    s->tau[i] = (-alphaCoeff2) * powf(c0, alphaPower - 1.0f);
    s->eta[i] = alphaCoeff2 * powf(c0, alphaPower) * tanPi2AlphaPower;
  }
}

kwo_sim* kwo_create(const kwo_problem* prob)
{
  kwo_sim* s = (kwo_sim*)calloc(1, sizeof(kwo_sim));
  s->pr  = *prob;
  s->nx  = prob->nx; s->ny = prob->ny; s->nz = prob->nz;
  s->nxc = s->nx / 2 + 1;
  s->n   = s->nx * s->ny * s->nz;
  s->nc  = s->nxc * s->ny * s->nz;
  s->fft_divider = 1.0f / (float)s->n; /* CudaParameters.cpp:259 */
  const size_t n = s->n;
  s->p = falloc(n); s->ux = falloc(n); s->uy = falloc(n); s->uz = falloc(n);
  s->rhox = falloc(n); s->rhoy = falloc(n); s->rhoz = falloc(n);
  s->duxdx = falloc(n); s->duydy = falloc(n); s->duzdz = falloc(n);
  s->t1 = falloc(n); s->t2 = falloc(n); s->t3 = falloc(n);
  s->cx = (cpx*)calloc(s->nc, sizeof(cpx));
  s->cy = (cpx*)calloc(s->nc, sizeof(cpx));
  s->cz = (cpx*)calloc(s->nc, sizeof(cpx));
  s->kappa = falloc(s->nc);

  /* dt / rho0_sg (KSpaceFirstOrderSolver.cpp:825-830; BaseFloatMatrix.cpp:86-93; Parameters.h:486-521) */
  if (prob->rho0)
  {
    s->dtrho0sgx = falloc(n); s->dtrho0sgy = falloc(n); s->dtrho0sgz = falloc(n);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++)
    {
      s->dtrho0sgx[i] = prob->dt / prob->rho0_sgx[i];
      s->dtrho0sgy[i] = prob->dt / prob->rho0_sgy[i];
      s->dtrho0sgz[i] = prob->dt / prob->rho0_sgz[i];
    }
    if (prob->dxudxn_sgx)
    { /* non-uniform grid: generateInitialDenisty (KSpaceFirstOrderSolver.cpp:2650-2685): (dt * dxudxn_sgx[x]) / rho0_sgx */
#pragma omp parallel for schedule(static)
      for (size_t z = 0; z < s->nz; z++)
        for (size_t y = 0; y < s->ny; y++)
          for (size_t x = 0; x < s->nx; x++)
          {
            const size_t i = (z * s->ny + y) * s->nx + x;
            s->dtrho0sgx[i] = (prob->dt * prob->dxudxn_sgx[x]) / prob->rho0_sgx[i];
            s->dtrho0sgy[i] = (prob->dt * prob->dyudyn_sgy[y]) / prob->rho0_sgy[i];
            s->dtrho0sgz[i] = (prob->dt * prob->dzudzn_sgz[z]) / prob->rho0_sgz[i];
          }
    }
  }
  else
  {
    s->dtrho0sgx_s = prob->dt / prob->rho0_sgx_s;
    s->dtrho0sgy_s = prob->dt / prob->rho0_sgy_s;
    s->dtrho0sgz_s = prob->dt / prob->rho0_sgz_s;
    s->dtrho0_s    = prob->rho0_s * prob->dt; /* CudaParameters.cpp:268 */
  }
  if (prob->absorbing_flag)
  {
    s->nabla1 = falloc(s->nc);
    s->nabla2 = falloc(s->nc);
    generate_kappa_and_nablas(s);
    generate_tau_eta(s);
  }
  else
  {
    generate_kappa(s, 0);
  }
  const int any_src = prob->p_source_flag || prob->ux_source_flag || prob->uy_source_flag || prob->uz_source_flag;
  if ((prob->u_source_mode == KWO_SRC_ADDITIVE || prob->p_source_mode == KWO_SRC_ADDITIVE) && any_src)
  {
    s->source_kappa = falloc(s->nc);
    generate_kappa(s, 1);
  }
  /* c^2 (KSpaceFirstOrderSolver.cpp:2690-2703; Parameters.h:453-456) */
  if (prob->c0)
  {
    s->c2 = falloc(n);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) s->c2[i] = prob->c0[i] * prob->c0[i];
  }
  else
  {
    s->c2_s = prob->c0_s * prob->c0_s;
  }
  return s;
}

void kwo_destroy(kwo_sim* s)
{
  if (!s) return;
  float* fl[] = { s->p, s->ux, s->uy, s->uz, s->rhox, s->rhoy, s->rhoz, s->duxdx, s->duydy, s->duzdz, s->t1, s->t2,
                  s->t3, s->c2, s->dtrho0sgx, s->dtrho0sgy, s->dtrho0sgz, s->tau, s->eta, s->kappa, s->nabla1,
                  s->nabla2, s->source_kappa };
  for (size_t i = 0; i < sizeof(fl) / sizeof(fl[0]); i++) free(fl[i]);
  free(s->cx); free(s->cy); free(s->cz);
  free(s);
}

uint64_t kwo_time_index(const kwo_sim* s) { return s->t; }

float* kwo_field(kwo_sim* s, const char* name)
{
#define F(nm, ptr) if (!strcmp(name, nm)) return (float*)(ptr)
  F("p", s->p); F("ux", s->ux); F("uy", s->uy); F("uz", s->uz);
  F("rhox", s->rhox); F("rhoy", s->rhoy); F("rhoz", s->rhoz);
  F("duxdx", s->duxdx); F("duydy", s->duydy); F("duzdz", s->duzdz);
  F("kappa", s->kappa); F("nabla1", s->nabla1); F("nabla2", s->nabla2); F("source_kappa", s->source_kappa);
  F("tau", s->tau); F("eta", s->eta); F("c2", s->c2);
  F("dtrho0sgx", s->dtrho0sgx); F("dtrho0sgy", s->dtrho0sgy); F("dtrho0sgz", s->dtrho0sgz);
  F("temp1", s->t1); F("temp2", s->t2); F("temp3", s->t3);
#undef F
  return NULL;
}
float kwo_scalar(const kwo_sim* s, const char* name)
{
#define F(nm, v) if (!strcmp(name, nm)) return (v)
  F("tau", s->tau_s); F("eta", s->eta_s); F("c2", s->c2_s);
  F("dtrho0sgx", s->dtrho0sgx_s); F("dtrho0sgy", s->dtrho0sgy_s); F("dtrho0sgz", s->dtrho0sgz_s);
  F("dtrho0", s->dtrho0_s); F("fft_divider", s->fft_divider);
#undef F
  return NAN;
}

/* ------------------------------------------------------------------------------------------------
 * The step
 * ---------------------------------------------------------------------------------------------- */

/* A2: SolverCudaKernels.cu:1139-1157 */
static void pressure_gradient(kwo_sim* s)
{
  const cpx* ddx = (const cpx*)s->pr.ddx_k_shift_pos;
  const cpx* ddy = (const cpx*)s->pr.ddy_k_shift_pos;
  const cpx* ddz = (const cpx*)s->pr.ddz_k_shift_pos;
#pragma omp parallel for schedule(static)
  for (size_t z = 0; z < s->nz; z++)
    for (size_t y = 0; y < s->ny; y++)
      for (size_t x = 0; x < s->nxc; x++)
      {
        const size_t i = (z * s->ny + y) * s->nxc + x;
        const cpx e = cscale(s->cx[i], s->kappa[i]);
        s->cx[i] = cmul(e, ddx[x]);
        s->cy[i] = cmul(e, ddy[y]);
        s->cz[i] = cmul(e, ddz[z]);
      }
}

/* A4: SolverCudaKernels.cu:184-215 (heterogeneous), :278-308 (homogeneous uniform) */
static void velocity_update(kwo_sim* s)
{
  const float d = s->fft_divider;
  const float *px = s->pr.pml_x_sgx, *py = s->pr.pml_y_sgy, *pz = s->pr.pml_z_sgz;
  const int het = s->dtrho0sgx != NULL;
  const float divX = s->dtrho0sgx_s * d, divY = s->dtrho0sgy_s * d, divZ = s->dtrho0sgz_s * d;
#pragma omp parallel for schedule(static)
  for (size_t z = 0; z < s->nz; z++)
    for (size_t y = 0; y < s->ny; y++)
      for (size_t x = 0; x < s->nx; x++)
      {
        const size_t i = (z * s->ny + y) * s->nx + x;
        const float ex = px[x], ey = py[y], ez = pz[z];
        if (het)
        {
          const float gx = d * s->t1[i] * s->dtrho0sgx[i];
          const float gy = d * s->t2[i] * s->dtrho0sgy[i];
          const float gz = d * s->t3[i] * s->dtrho0sgz[i];
          s->ux[i] = (s->ux[i] * ex - gx) * ex;
          s->uy[i] = (s->uy[i] * ey - gy) * ey;
          s->uz[i] = (s->uz[i] * ez - gz) * ez;
        }
        else if (s->pr.dxudxn_sgx)
        { /* homogeneous, non-uniform grid: SolverCudaKernels.cu:372-410 */
          const float gx = divX * s->pr.dxudxn_sgx[x] * s->t1[i];
          const float gy = divY * s->pr.dyudyn_sgy[y] * s->t2[i];
          const float gz = divZ * s->pr.dzudzn_sgz[z] * s->t3[i];
          s->ux[i] = (s->ux[i] * ex - gx) * ex;
          s->uy[i] = (s->uy[i] * ey - gy) * ey;
          s->uz[i] = (s->uz[i] * ez - gz) * ez;
        }
        else
        {
          s->ux[i] = (s->ux[i] * ex - divX * s->t1[i]) * ex;
          s->uy[i] = (s->uy[i] * ey - divY * s->t2[i]) * ey;
          s->uz[i] = (s->uz[i] * ez - divZ * s->t3[i]) * ez;
        }
      }
}

/* scaleSource: KSpaceFirstOrderSolver.cpp:2339-2352; SolverCudaKernels.cu:679-697,740-745 */
static void scale_source(kwo_sim* s, float* scaled, const float* input, const uint64_t* index, size_t nsrc, int many)
{
  memset(scaled, 0, sizeof(float) * s->n);
  const size_t index2D = many ? s->t * nsrc : s->t;
  for (size_t i = 0; i < nsrc; i++) scaled[index[i]] = many ? input[index2D + i] : input[index2D];
  kwo_fft_r2c_3d(scaled, (float*)s->cx, s->nx, s->ny, s->nz);
  const float d = s->fft_divider;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < s->nc; i++) s->cx[i] = cscale(s->cx[i], s->source_kappa[i] * d);
  kwo_fft_c2r_3d((const float*)s->cx, scaled, s->nx, s->ny, s->nz);
}

/* A5: SolverCudaKernels.cu:504-528; KSpaceFirstOrderSolver.cpp:2252-2303 */
static void add_velocity_source_one(kwo_sim* s, float* u, const float* input, uint64_t flag)
{
  const kwo_problem* pr = &s->pr;
  if (!(flag > s->t)) return;
  const size_t n = pr->u_source_n;
  if (pr->u_source_mode != KWO_SRC_ADDITIVE)
  {
    const size_t index2D = (pr->u_source_many == 0) ? s->t : s->t * n;
    for (size_t i = 0; i < n; i++)
    {
      const float v = (pr->u_source_many == 0) ? input[index2D] : input[index2D + i];
      if (pr->u_source_mode == KWO_SRC_DIRICHLET) u[pr->u_source_index[i]] = v;
      else u[pr->u_source_index[i]] += v;
    }
  }
  else
  {
    scale_source(s, s->t1, input, pr->u_source_index, n, pr->u_source_many);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < s->n; i++) u[i] += s->t1[i]; /* SolverCudaKernels.cu:765-770 */
  }
}

/* A7: SolverCudaKernels.cu:1210-1239 */
static void velocity_gradient(kwo_sim* s)
{
  const cpx* ddx = (const cpx*)s->pr.ddx_k_shift_neg;
  const cpx* ddy = (const cpx*)s->pr.ddy_k_shift_neg;
  const cpx* ddz = (const cpx*)s->pr.ddz_k_shift_neg;
  const float d = s->fft_divider;
#pragma omp parallel for schedule(static)
  for (size_t z = 0; z < s->nz; z++)
    for (size_t y = 0; y < s->ny; y++)
      for (size_t x = 0; x < s->nxc; x++)
      {
        const size_t i = (z * s->ny + y) * s->nxc + x;
        const float ek = s->kappa[i] * d;
        s->cx[i] = cmul(cscale(s->cx[i], ek), ddx[x]);
        s->cy[i] = cmul(cscale(s->cy[i], ek), ddy[y]);
        s->cz[i] = cmul(cscale(s->cz[i], ek), ddz[z]);
      }
}

/* A9: SolverCudaKernels.cu:1358-1393 (nonlinear), :1470-1497 (linear) */
static void density_update(kwo_sim* s)
{
  const kwo_problem* pr = &s->pr;
  const float *px = pr->pml_x, *py = pr->pml_y, *pz = pr->pml_z;
  const float dt = pr->dt;
#pragma omp parallel for schedule(static)
  for (size_t z = 0; z < s->nz; z++)
    for (size_t y = 0; y < s->ny; y++)
      for (size_t x = 0; x < s->nx; x++)
      {
        const size_t i = (z * s->ny + y) * s->nx + x;
        const float ex = px[x], ey = py[y], ez = pz[z];
        if (pr->nonlinear_flag)
        {
          const float rx = s->rhox[i], ry = s->rhoy[i], rz = s->rhoz[i];
          const float r0 = pr->rho0 ? pr->rho0[i] : pr->rho0_s;
          const float sumRhosDt = (2.0f * (rx + ry + rz) + r0) * dt;
          s->rhox[i] = ex * ((ex * rx) - sumRhosDt * s->duxdx[i]);
          s->rhoy[i] = ey * ((ey * ry) - sumRhosDt * s->duydy[i]);
          s->rhoz[i] = ez * ((ez * rz) - sumRhosDt * s->duzdz[i]);
        }
        else
        {
          const float dtRho0 = pr->rho0 ? dt * pr->rho0[i] : s->dtrho0_s;
          s->rhox[i] = ex * (ex * s->rhox[i] - dtRho0 * s->duxdx[i]);
          s->rhoy[i] = ey * (ey * s->rhoy[i] - dtRho0 * s->duydy[i]);
          s->rhoz[i] = ez * (ez * s->rhoz[i] - dtRho0 * s->duzdz[i]);
        }
      }
}

/* A10: SolverCudaKernels.cu:570-629, :795-807; KSpaceFirstOrderSolver.cpp:2310-2332 */
static void add_pressure_source(kwo_sim* s)
{
  const kwo_problem* pr = &s->pr;
  if (!(pr->p_source_flag > s->t)) return;
  const size_t n = pr->p_source_n;
  const int is3d = (s->nz != 1); /* 2-D (SD::k2D instantiations): the source goes to rho_x and rho_y only */
  if (pr->p_source_mode != KWO_SRC_ADDITIVE)
  {
    const size_t index2D = (pr->p_source_many == 0) ? s->t : s->t * n;
    for (size_t i = 0; i < n; i++)
    {
      const float v = (pr->p_source_many == 0) ? pr->p_source_input[index2D] : pr->p_source_input[index2D + i];
      const size_t j = pr->p_source_index[i];
      if (pr->p_source_mode == KWO_SRC_DIRICHLET) { s->rhox[j] = v; s->rhoy[j] = v; if (is3d) s->rhoz[j] = v; }
      else { s->rhox[j] += v; s->rhoy[j] += v; if (is3d) s->rhoz[j] += v; }
    }
  }
  else
  {
    scale_source(s, s->t1, pr->p_source_input, pr->p_source_index, n, pr->p_source_many);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < s->n; i++)
    {
      const float e = s->t1[i];
      s->rhox[i] += e; s->rhoy[i] += e; if (is3d) s->rhoz[i] += e;
    }
  }
}

/* A11: KSpaceFirstOrderSolver.cpp:2180-2245 and the kernels cited per branch */
static void pressure_update(kwo_sim* s)
{
  const kwo_problem* pr = &s->pr;
  const float d = s->fft_divider;
  const size_t n = s->n;
  if (!pr->absorbing_flag)
  {
    if (!pr->nonlinear_flag)
    { /* SolverCudaKernels.cu:2224-2236 */
#pragma omp parallel for schedule(static)
      for (size_t i = 0; i < n; i++)
      {
        const float c2 = s->c2 ? s->c2[i] : s->c2_s;
        const float sum = s->rhox[i] + s->rhoy[i] + s->rhoz[i];
        s->p[i] = c2 * sum;
      }
    }
    else
    { /* SolverCudaKernels.cu:2067-2084 */
#pragma omp parallel for schedule(static)
      for (size_t i = 0; i < n; i++)
      {
        const float c2 = s->c2 ? s->c2[i] : s->c2_s;
        const float bona = pr->bona ? pr->bona[i] : pr->bona_s;
        const float r0 = pr->rho0 ? pr->rho0[i] : pr->rho0_s;
        const float rs = s->rhox[i] + s->rhoy[i] + s->rhoz[i];
        s->p[i] = c2 * (rs + (bona * (rs * rs) / (2.0f * r0)));
      }
    }
    return;
  }
  if (pr->nonlinear_flag)
  {
    /* terms: SolverCudaKernels.cu:1577-1602; aliasing KSpaceFirstOrderSolver.cpp:2184-2190 */
    float *densitySum = s->t1, *nonlinearTerm = s->t2, *velGradSum = s->t3;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++)
    {
      const float bona = pr->bona ? pr->bona[i] : pr->bona_s;
      const float r0 = pr->rho0 ? pr->rho0[i] : pr->rho0_s;
      const float rs = s->rhox[i] + s->rhoy[i] + s->rhoz[i];
      const float ds = s->duxdx[i] + s->duydy[i] + s->duzdz[i];
      densitySum[i] = rs;
      nonlinearTerm[i] = ((bona * rs * rs) / (2.0f * r0)) + rs;
      velGradSum[i] = r0 * ds;
    }
    kwo_fft_r2c_3d(velGradSum, (float*)s->cx, s->nx, s->ny, s->nz);
    kwo_fft_r2c_3d(densitySum, (float*)s->cy, s->nx, s->ny, s->nz);
    /* SolverCudaKernels.cu:1812-1820 */
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < s->nc; i++)
    {
      s->cx[i] = cscale(s->cx[i], s->nabla1[i]);
      s->cy[i] = cscale(s->cy[i], s->nabla2[i]);
    }
    float *tauTerm = velGradSum, *etaTerm = densitySum;
    kwo_fft_c2r_3d((const float*)s->cx, tauTerm, s->nx, s->ny, s->nz);
    kwo_fft_c2r_3d((const float*)s->cy, etaTerm, s->nx, s->ny, s->nz);
    /* SolverCudaKernels.cu:1865-1879 */
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++)
    {
      const float c2 = s->c2 ? s->c2[i] : s->c2_s;
      const float tau = s->tau ? s->tau[i] : s->tau_s;
      const float eta = s->eta ? s->eta[i] : s->eta_s;
      s->p[i] = c2 * (nonlinearTerm[i] + (d * ((tauTerm[i] * tau) - (etaTerm[i] * eta))));
    }
  }
  else
  {
    /* terms: SolverCudaKernels.cu:1724-1742; aliasing KSpaceFirstOrderSolver.cpp:2221-2225 */
    float *densitySum = s->t1, *velGradTerm = s->t2;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++)
    {
      const float r0 = pr->rho0 ? pr->rho0[i] : pr->rho0_s;
      densitySum[i] = s->rhox[i] + s->rhoy[i] + s->rhoz[i];
      const float ds = s->duxdx[i] + s->duydy[i] + s->duzdz[i];
      velGradTerm[i] = r0 * ds;
    }
    kwo_fft_r2c_3d(velGradTerm, (float*)s->cx, s->nx, s->ny, s->nz);
    kwo_fft_r2c_3d(densitySum, (float*)s->cy, s->nx, s->ny, s->nz);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < s->nc; i++)
    {
      s->cx[i] = cscale(s->cx[i], s->nabla1[i]);
      s->cy[i] = cscale(s->cy[i], s->nabla2[i]);
    }
    float *tauTerm = s->t2, *etaTerm = s->t3;
    kwo_fft_c2r_3d((const float*)s->cx, tauTerm, s->nx, s->ny, s->nz);
    kwo_fft_c2r_3d((const float*)s->cy, etaTerm, s->nx, s->ny, s->nz);
    /* SolverCudaKernels.cu:1966-1980 */
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++)
    {
      const float c2 = s->c2 ? s->c2[i] : s->c2_s;
      const float tau = s->tau ? s->tau[i] : s->tau_s;
      const float eta = s->eta ? s->eta[i] : s->eta_s;
      s->p[i] = c2 * (densitySum[i] + (d * (tauTerm[i] * tau - etaTerm[i] * eta)));
    }
  }
}

/* A12: KSpaceFirstOrderSolver.cpp:2359-2396; SolverCudaKernels.cu:864-884, :949-982 */
static void add_initial_pressure_source(kwo_sim* s)
{
  const kwo_problem* pr = &s->pr;
  const size_t n = s->n;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++)
  {
    float tmp = s->p[i] = pr->p0_source_input[i];
    const float c2 = s->c2 ? s->c2[i] : s->c2_s;
    /* SolverCudaKernels.cu:873-876: dimScalingFactor = 3 (k3D) or 2 (k2D, Nz == 1; rho_z does not exist there) */
    const float dims = (s->nz == 1) ? 2.0f : 3.0f;
    tmp = tmp / (dims * c2);
    s->rhox[i] = tmp; s->rhoy[i] = tmp; s->rhoz[i] = (s->nz == 1) ? 0.0f : tmp;
  }
  kwo_fft_r2c_3d(s->p, (float*)s->cx, s->nx, s->ny, s->nz);
  pressure_gradient(s);
  kwo_fft_c2r_3d((const float*)s->cx, s->ux, s->nx, s->ny, s->nz);
  kwo_fft_c2r_3d((const float*)s->cy, s->uy, s->nx, s->ny, s->nz);
  kwo_fft_c2r_3d((const float*)s->cz, s->uz, s->nx, s->ny, s->nz);
  const float d = s->fft_divider;
  if (s->dtrho0sgx)
  {
    const float divider = d * 0.5f;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++)
    {
      s->ux[i] *= s->dtrho0sgx[i] * divider;
      s->uy[i] *= s->dtrho0sgy[i] * divider;
      s->uz[i] *= s->dtrho0sgz[i] * divider;
    }
  }
  else
  {
    const float dX = d * 0.5f * s->dtrho0sgx_s, dY = d * 0.5f * s->dtrho0sgy_s, dZ = d * 0.5f * s->dtrho0sgz_s;
    if (pr->dxudxn_sgx)
    { /* computeInitialVelocityHomogeneousNonuniform (SolverCudaKernels.cu:1061-1083) */
#pragma omp parallel for schedule(static)
      for (size_t z = 0; z < s->nz; z++)
        for (size_t y = 0; y < s->ny; y++)
          for (size_t x = 0; x < s->nx; x++)
          {
            const size_t i = (z * s->ny + y) * s->nx + x;
            s->ux[i] *= dX * pr->dxudxn_sgx[x];
            s->uy[i] *= dY * pr->dyudyn_sgy[y];
            s->uz[i] *= dZ * pr->dzudzn_sgz[z];
          }
    }
    else
    {
#pragma omp parallel for schedule(static)
      for (size_t i = 0; i < n; i++) { s->ux[i] *= dX; s->uy[i] *= dY; s->uz[i] *= dZ; }
    }
  }
}

/* KSpaceFirstOrderSolver.cpp:885-935 (without storeSensorData/printStatistics) */
void kwo_step(kwo_sim* s)
{
  const kwo_problem* pr = &s->pr;
  /* computeVelocity :2087-2119 */
  kwo_fft_r2c_3d(s->p, (float*)s->cx, s->nx, s->ny, s->nz);
  pressure_gradient(s);
  kwo_fft_c2r_3d((const float*)s->cx, s->t1, s->nx, s->ny, s->nz);
  kwo_fft_c2r_3d((const float*)s->cy, s->t2, s->nx, s->ny, s->nz);
  kwo_fft_c2r_3d((const float*)s->cz, s->t3, s->nx, s->ny, s->nz);
  velocity_update(s);
  /* addVelocitySource :2252-2303 */
  add_velocity_source_one(s, s->ux, pr->ux_source_input, pr->ux_source_flag);
  add_velocity_source_one(s, s->uy, pr->uy_source_input, pr->uy_source_flag);
  add_velocity_source_one(s, s->uz, pr->uz_source_input, pr->uz_source_flag);
  /* transducer :894-897; SolverCudaKernels.cu:463-471 */
  if (pr->transducer_source_flag > s->t)
    for (size_t i = 0; i < pr->u_source_n; i++)
      s->ux[pr->u_source_index[i]] += pr->transducer_source_input[pr->delay_mask[i] + s->t];
  /* computeVelocityGradient :2126-2150 */
  kwo_fft_r2c_3d(s->ux, (float*)s->cx, s->nx, s->ny, s->nz);
  kwo_fft_r2c_3d(s->uy, (float*)s->cy, s->nx, s->ny, s->nz);
  kwo_fft_r2c_3d(s->uz, (float*)s->cz, s->nx, s->ny, s->nz);
  velocity_gradient(s);
  kwo_fft_c2r_3d((const float*)s->cx, s->duxdx, s->nx, s->ny, s->nz);
  kwo_fft_c2r_3d((const float*)s->cy, s->duydy, s->nx, s->ny, s->nz);
  kwo_fft_c2r_3d((const float*)s->cz, s->duzdz, s->nx, s->ny, s->nz);
  if (pr->dxudxn)
  { /* non-uniform grid: computeVelocityGradientShiftNonuniform (SolverCudaKernels.cu:1285-1301; …Solver.cpp:2145-2149) */
#pragma omp parallel for schedule(static)
    for (size_t z = 0; z < s->nz; z++)
      for (size_t y = 0; y < s->ny; y++)
        for (size_t x = 0; x < s->nx; x++)
        {
          const size_t i = (z * s->ny + y) * s->nx + x;
          s->duxdx[i] *= pr->dxudxn[x];
          s->duydy[i] *= pr->dyudyn[y];
          s->duzdz[i] *= pr->dzudzn[z];
        }
  }
  density_update(s);
  add_pressure_source(s);
  pressure_update(s);
  if (s->t == 0 && pr->p0_source_flag == 1) add_initial_pressure_source(s);
  s->t++;
}

/* ------------------------------------------------------------------------------------------------
 * Sampling: OutputStreams/OutputStreamsCudaKernels.cu
 * ---------------------------------------------------------------------------------------------- */
static inline void reduce_one(int op, float* b, float v)
{
  switch (op)
  {
    case KWO_OP_NONE: *b = v; break;
    case KWO_OP_RMS: *b = fmaf(v, v, *b); break; /* nvcc default -fmad=true contracts buf += v*v (:91-93) into one FMA */
    case KWO_OP_MAX: *b = (*b > v) ? *b : v; break; /* max(buf, v): :97-99 */
    case KWO_OP_MIN: *b = (*b < v) ? *b : v; break;
  }
}
void kwo_sample_index(int op, float* buf, const float* src, const uint64_t* mask, uint64_t n)
{ /* :83-107 */
  for (uint64_t i = 0; i < n; i++) reduce_one(op, &buf[i], src[mask[i]]);
}
void kwo_sample_cuboid(int op, float* buf, const float* src, const uint32_t tl[3], const uint32_t br[3],
                       const uint32_t size[3], uint64_t n)
{ /* :164-230 */
  const uint64_t cx = br[0] - tl[0] + 1, cy = br[1] - tl[1] + 1;
  const uint64_t slab = cx * cy;
  for (uint64_t i = 0; i < n; i++)
  {
    const uint64_t lz = i / slab, ly = (i % slab) / cx, lx = (i % slab) % cx;
    const uint64_t pos = (lz + tl[2]) * (uint64_t)size[0] * size[1] + (ly + tl[1]) * (uint64_t)size[0] + (lx + tl[0]);
    reduce_one(op, &buf[i], src[pos]);
  }
}
void kwo_sample_all(int op, float* buf, const float* src, uint64_t n)
{ /* :297-316 */
  for (uint64_t i = 0; i < n; i++) reduce_one(op, &buf[i], src[i]);
}
void kwo_post_rms(float* buf, float scale, uint64_t n)
{ /* :359-365 */
  for (uint64_t i = 0; i < n; i++) buf[i] = sqrtf(buf[i] * scale);
}

/* ------------------------------------------------------------------------------------------------
 * Shifted velocity: KSpaceFirstOrderSolver.cpp:2714-2735; SolverCudaKernels.cu:2617-2689
 * ---------------------------------------------------------------------------------------------- */
void kwo_shifted_velocity(const float* u, float* out, const float* shift_neg_r, uint64_t nx, uint64_t ny, uint64_t nz,
                          int axis)
{
  const size_t n_axis = (axis == 0) ? nx : (axis == 1) ? ny : nz;
  const size_t nr = n_axis / 2 + 1;
  const size_t ox = (axis == 0) ? nr : nx, oy = (axis == 1) ? nr : ny, oz = (axis == 2) ? nr : nz;
  cpx* tmp = (cpx*)malloc(sizeof(cpx) * ox * oy * oz);
  kwo_fft_r2c_1d(u, (float*)tmp, nx, ny, nz, axis);
  const cpx* sh = (const cpx*)shift_neg_r;
  const float divider = 1.0f / (float)n_axis; /* CudaParameters.cpp:260-262 */
#pragma omp parallel for schedule(static)
  for (size_t z = 0; z < oz; z++)
    for (size_t y = 0; y < oy; y++)
      for (size_t x = 0; x < ox; x++)
      {
        const size_t i = (z * oy + y) * ox + x;
        const size_t k = (axis == 0) ? x : (axis == 1) ? y : z;
        tmp[i] = cscale(cmul(tmp[i], sh[k]), divider);
      }
  kwo_fft_c2r_1d((const float*)tmp, out, nx, ny, nz, axis);
  free(tmp);
}

/* ------------------------------------------------------------------------------------------------
 * Compression: Compression/CompressHelper.cpp:48-65,672-778; OutputStreams/IndexOutputStream.cpp:373-470
 * ---------------------------------------------------------------------------------------------- */
uint64_t kwo_compress_osize(float period, uint64_t mos) { return (uint64_t)(period * (float)mos); }
uint64_t kwo_compress_bsize(float period, uint64_t mos) { return kwo_compress_osize(period, mos) * 2 + 1; }

void kwo_compress_basis(float period, uint64_t mos, uint64_t harmonics, int shifted, float* bE_, float* bE_1_)
{
  const uint64_t o = kwo_compress_osize(period, mos), bs = 2 * o + 1;
  cpx* bE = (cpx*)bE_;
  cpx* bE_1 = (cpx*)bE_1_;
  float* b = (float*)malloc(sizeof(float) * bs);
  cpx* e = (cpx*)malloc(sizeof(cpx) * bs);
  for (uint64_t x = 0; x < o; x++) b[x] = (float)x / (float)o;
  for (uint64_t x = o; x < 2 * o + 1; x++) b[x] = 2.0f - (float)x / (float)o;
  for (uint64_t ih = 0; ih < harmonics; ih++)
  {
    const float h = (float)(ih + 1);
    for (uint64_t x = 0; x < bs; x++)
    {
      /* e = exp(-i * (2 pi / (period / h)) * x) [* exp(+i pi / (period / h))] */
      const float w = 2.0f * (float)M_PI / (period / h);
      const float ph = -(w * (float)x);
      cpx v = { cosf(ph), sinf(ph) };
      if (shifted)
      {
        const float ps = (float)M_PI / (period / h);
        cpx sft = { cosf(ps), sinf(ps) };
        v = cmul(v, sft);
      }
      e[x] = v;
    }
    for (uint64_t x = 0; x < bs; x++)
    {
      const uint64_t hx = ih * bs + x;
      const uint64_t xo = (x + o) % (bs - 1);
      bE[hx] = cscale(e[x], b[x]);
      bE_1[hx] = cscale(e[xo], b[xo]);
      bE[hx] = cscale(bE[hx], 2.0f / (float)o);
      bE_1[hx] = cscale(bE_1[hx], 2.0f / (float)o);
    }
  }
  free(b);
  free(e);
}

int kwo_compress_step(kwo_compress_state* st, const float* bE_, const float* bE_1_, const float* x, int is_last_step,
                      float* frame_out)
{
  const cpx* bE = (const cpx*)bE_;
  const cpx* bE_1 = (const cpx*)bE_1_;
  cpx* c1 = (cpx*)st->c1;
  cpx* c2 = (cpx*)st->c2;
  const uint64_t step_local = st->sampled_step % (st->b_size - 1);
  const int saving = ((step_local + 1) % st->o_size == 0);
  const int odd = ((st->compressed_step + 1) % 2 == 0);
  const int mirror = (st->compressed_step == 0 && saving && !st->no_overlap);
  for (uint64_t i = 0; i < st->n_sens; i++)
    for (uint64_t ih = 0; ih < st->harmonics; ih++)
    {
      const uint64_t ph = st->harmonics * i + ih;
      const uint64_t bi = ih * st->b_size + step_local;
      c1[ph] = cadd(c1[ph], cscale(bE[bi], x[i]));
      c2[ph] = cadd(c2[ph], cscale(bE_1[bi], x[i]));
      if (mirror) c2[ph] = cadd(c2[ph], c1[ph]);
    }
  int emitted = 0;
  if (saving || is_last_step)
  {
    cpx* cur = odd ? c1 : c2;
    memcpy(frame_out, cur, sizeof(cpx) * st->n_sens * st->harmonics);
    st->compressed_step++;
    emitted = 1;
    if (saving) memset(cur, 0, sizeof(cpx) * st->n_sens * st->harmonics); /* BaseOutputStream.cpp:117-132 */
  }
  st->sampled_step++;
  return emitted;
}
