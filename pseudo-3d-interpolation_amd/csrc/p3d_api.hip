// p3d_api.hip -- C ABI (include/p3d.h) over the HIP kernels: plan management, schedule statistics,
// the POCS iteration driver, staging for host-pointer callers.
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "p3d.h"
#include "p3d_generic.hpp"
#include "p3d_internal.hpp"
#include "p3d_flex.hpp"
#include "p3d_mix_entry.hpp"
#include "p3d_resident.hpp"
#include "p3d_select.hpp"
#include "p3d_kernels.hpp"

namespace p3d {
#define P3D_DECL(n) const LineOps* get_line_ops_##n();
P3D_DECL(2) P3D_DECL(4) P3D_DECL(8) P3D_DECL(16) P3D_DECL(32) P3D_DECL(64) P3D_DECL(128) P3D_DECL(256)
P3D_DECL(512) P3D_DECL(1024) P3D_DECL(2048) P3D_DECL(4096)
#undef P3D_DECL

static const LineOps* find_ops(int n)
{
    switch (n) {
        case 2: return get_line_ops_2();
        case 4: return get_line_ops_4();
        case 8: return get_line_ops_8();
        case 16: return get_line_ops_16();
        case 32: return get_line_ops_32();
        case 64: return get_line_ops_64();
        case 128: return get_line_ops_128();
        case 256: return get_line_ops_256();
        case 512: return get_line_ops_512();
        case 1024: return get_line_ops_1024();
        case 2048: return get_line_ops_2048();
        case 4096: return get_line_ops_4096();
        default: break;
    }
    // any other length whose lines fit LDS: same two passes, LDS-resident mixed-radix transforms (p3d_flex.hip)
    if (flex_supported(n) && !getenv("P3D_NO_FLEX")) return get_flex_ops();
    return nullptr;
}
static bool is_flex(const LineOps* ops) { return ops && ops->tpl == 0; }
// largest slice the single-kernel path takes: 128 x 128 needs a 1024-thread workgroup, i.e. 128 registers per thread, and the
// column transform through a 136-KiB LDS image does not fit them (350 B of scratch per thread) -- it stays with the two passes
static const size_t RESIDENT_MAX_POINTS = 8192;
}  // namespace p3d

using namespace p3d;

// ---- rowbase[r] = number of observed positions in rows < r (rowbase[n1] = total), from the packed mask ---------
static __global__ void rowbase_kernel(const uint16_t* bits, unsigned* rowbase, int n1, int tpl)
{
    __shared__ unsigned cnt[4096 + 1];
    for (int r = threadIdx.x; r < n1; r += blockDim.x) {
        unsigned c = 0;
        for (int t = 0; t < tpl; ++t) c += __popc((unsigned)bits[(size_t)r * tpl + t]);
        cnt[r] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned run = 0;
        for (int r = 0; r < n1; ++r) { const unsigned c = cnt[r]; cnt[r] = run; run += c; }
        cnt[n1] = run;
    }
    __syncthreads();
    for (int r = threadIdx.x; r <= n1; r += blockDim.x) rowbase[r] = cnt[r];
}


// Column-pass tile flags -> one 16-bit word per (slice, group of 8 row-pass threads): bit q = the column block that holds
// element tl + tpl*q kept a coefficient.  col_t = columns per column-pass tile (a block spans 8/col_t tiles or a tile spans
// col_t/8 blocks).  Also counts the kept blocks (statistics only).
static __global__ void nz_count_kernel(const uint8_t* flags, unsigned long long* count, int nslices, int tiles, int col_t, int nblocks, const int* done)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < nslices * nblocks;
    const int s = live ? i / nblocks : 0, b = live ? i - s * nblocks : 0;
    unsigned any = 0;
    if (live && !(done && done[s] != 0)) {
        const uint8_t* f = flags + (size_t)s * tiles;
        if (col_t >= 8) any = f[b / (col_t / 8)];
        else for (int t = 0; t < 8 / col_t; ++t) { const int ti = b * (8 / col_t) + t; if (ti < tiles) any |= f[ti]; }
    }
    const unsigned long long kept = __ballot(any != 0);   // one atomic per wavefront, not one per kept block (all on one address)
    if ((threadIdx.x & 63u) == 0 && kept) atomicAdd(count, (unsigned long long)__popcll(kept));
}

static __global__ void nz_pack_kernel(const uint8_t* flags, uint16_t* nzm, unsigned long long* count, int nslices, int tiles, int col_t, int groups,
                                      int nblocks, const int* done, unsigned long long* nzl)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < nslices * groups;
    const int s = live ? i / groups : 0, g = live ? i - s * groups : 0;
    const bool skip = !live || (done && done[s] != 0);   // finished / empty slices: nobody reads their word
    unsigned word = 0, kept = 0;
    if (!skip) {
        const uint8_t* f = flags + (size_t)s * tiles;
        for (int q = 0; q < 16; ++q) {
            const int b = g + groups * q;
            if (b >= nblocks) continue;
            unsigned any = 0;
            if (col_t >= 8) any = f[b / (col_t / 8)];
            else for (int t = 0; t < 8 / col_t; ++t) { const int ti = b * (8 / col_t) + t; if (ti < tiles) any |= f[ti]; }
            word |= (any ? 1u : 0u) << q;
            kept += any ? 1u : 0u;
        }
        nzm[i] = (uint16_t)word;
    }
    {   // kept blocks of the launch: one atomic per wavefront (one per thread -- a few thousand on one address -- was most of this kernel's 10 us)
        unsigned wsum = kept;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wsum += __shfl_down(wsum, o, 64);
        if ((threadIdx.x & 63u) == 0 && wsum) atomicAdd(count, (unsigned long long)wsum);
    }
    // rows of whole wavefronts (groups = 8, 16, 32): the same flags as 64-bit lane masks, word (slice, wavefront, q) bit l = the
    // flag word of thread group 8 * wavefront + l / 8, bit q; those 8 threads are neighbours in the wave and build two words each
    if (nzl != nullptr && groups < 8) {
        // rows shorter than a wavefront (tpl = 8 * groups lanes per row, 64 / tpl rows per wave): lane l belongs to thread group
        // (l % tpl) / 8 of its row; the `groups` threads of a slice are neighbours and build 16 / groups words each
        const int base = (int)(threadIdx.x & 63u) - g, tpl = 8 * groups, per = 16 / groups;
        unsigned long long pat[4] = {0, 0, 0, 0};
        unsigned wg[4] = {0, 0, 0, 0};
        for (int k = 0; k < groups; ++k) {
            wg[k] = (unsigned)__shfl((int)word, base + k, 64);
            for (int l0 = 0; l0 < 64; l0 += tpl) pat[k] |= 0xFFull << (l0 + 8 * k);
        }
        if (!skip) {
            for (int qq = 0; qq < per; ++qq) {
                const int q = g * per + qq;
                unsigned long long m = 0;
                for (int k = 0; k < groups; ++k) if ((wg[k] >> q) & 1u) m |= pat[k];
                nzl[(size_t)s * 16 + q] = m;
            }
        }
    } else if (nzl != nullptr) {
        const int base = (int)(threadIdx.x & 63u) & ~7;
        unsigned long long m0 = 0, m1 = 0;
        for (int k = 0; k < 8; ++k) {
            const unsigned wk = (unsigned)__shfl((int)word, base + k, 64);
            if ((wk >> (2 * (g % 8))) & 1u) m0 |= 0xFFull << (8 * k);
            if ((wk >> (2 * (g % 8) + 1)) & 1u) m1 |= 0xFFull << (8 * k);
        }
        if (!skip) {   // groups = 8 * (wavefronts per row); thread g serves wavefront g / 8, registers 2 * (g % 8) and + 1
            const size_t w0 = pipe64_word((size_t)s, groups / 8, g / 8, 2 * (g % 8));
            nzl[w0] = m0;
            nzl[w0 + 1] = m1;
        }
    }
}
// Real cubes (row_real_kernel): lane masks over the HALF spectrum.  Register q < 8 of lane l holds column l + 64 q, register
// q >= 8 reads the mirror column n2 - l - 64 q (n2 = 1024): word (slice, q) bit l = that column's 8-column block kept something.
static __global__ void nz_real_kernel(const uint8_t* flags, unsigned long long* nzl, unsigned long long* count, int nslices, int tiles, int col_t,
                                      int n2, int tpl, const int* done)
{
    // one wavefront per word: lane l decides its own bit, the word is the ballot
    const int wpl = tpl >= 64 ? tpl / 64 : 1;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
    if (i >= nslices * wpl * 16) return;
    const int q = i & 15, wsub = (i >> 4) % wpl, s = (i >> 4) / wpl;
    if (done && done[s] != 0) return;
    const uint8_t* f = flags + (size_t)s * tiles;
    auto block_kept = [&](int b) -> unsigned {
        unsigned any = 0;
        if (col_t >= 8) any = f[b / (col_t / 8)];
        else for (int t = 0; t < 8 / col_t; ++t) { const int ti = b * (8 / col_t) + t; if (ti < tiles) any |= f[ti]; }
        return any ? 1u : 0u;
    };
    const int e = (tpl >= 64 ? 64 * wsub + l : l % tpl) + tpl * q, k = q < 8 ? e : n2 - e;
    const unsigned long long w = __ballot(block_kept(k >> 3) != 0u);
    if (l == 0) nzl[i] = w;
    if (q == 0 && wsub == 0) {   // statistics: kept blocks of this slice's half spectrum
        unsigned kept = 0;
        for (int b = l; b <= (n2 / 2) >> 3; b += 64) kept += block_kept(b);
        for (int o = 32; o > 0; o >>= 1) kept += __shfl_down(kept, o, 64);
        if (l == 0 && kept) atomicAdd(count, (unsigned long long)kept);
    }
}

// ---- per-slice sum of the per-row sums, fixed order (bitwise reproducible) ----------------------------------
static __global__ void reduce_rows_kernel(const double* rowsum, double* sums_row, int n1)
{
    __shared__ double sh[256];
    const int s = blockIdx.x;
    double t = 0.0;
    for (int i = threadIdx.x; i < n1; i += 256) t += rowsum[(size_t)s * n1 + i];
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums_row[s] = sh[0];
}

// ---- debug (P3D_CHECK_DONE_ROWS=1): checksum of the work slice of every converged slice that still waits for its finalize launch ----
// The early exit hands a converged slice's iterate back from its WORK ROWS up to FIN_EVERY iterations later (p3d_pocs_run_dev), so every
// pass must leave the work slice of a slice with done != 0 alone.  sum[s] = wrap-around sum of the slice's 64-bit words, 0 for the others.
static __global__ void done_rows_checksum_kernel(const unsigned long long* work, size_t words_per_slice, const int* done, int lo, unsigned long long* sum)
{
    __shared__ unsigned long long sh[256];
    const int s = blockIdx.x;
    unsigned long long t = 0;
    if (done[s] > lo)
        for (size_t i = threadIdx.x; i < words_per_slice; i += 256) t += work[(size_t)s * words_per_slice + i] * (2 * i + 1);
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) sum[s] = sh[0];
}

// ---- sums[0] of slices that are switched off from the start (done < 0) reads as zero --------------------------
static __global__ void zero_off_sums_kernel(double* sums0, const int* done, int nslices)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nslices && done[s] != 0) sums0[s] = 0.0;
}

// ---- convergence bookkeeping (POCS.py:622, 631) ------------------------------------------------
static __global__ void conv_kernel(const double* sums, int* done, int nslices, int iter, double eps)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslices || done[s] != 0) return;
    const double cur = sums[(size_t)(iter + 1) * nslices + s];
    const double prev = sums[(size_t)iter * nslices + s];
    const double d = cur - prev;
    const double cost = (d * d) / (cur * cur);
    if (iter > 2 && cost < eps) done[s] = iter + 1;
}


// ---- packed trace mask ---------------------------------------------------------------------------
// bits[row][tl] bit q = (mask[row][tl + tpl*q] == 1); *nonbinary is raised when an entry is neither 0 nor 1
// (the reference accepts any mask with max <= 1, POCS.py:488; such masks take the float path).
static __global__ void pack_mask_kernel(const float* mask, uint16_t* bits, int* nonbinary, int n1, int n2, int tpl, int ppt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1 * tpl) return;
    const int row = i / tpl, tl = i - row * tpl;
    unsigned w = 0;
    bool odd = false;
    for (int q = 0; q < ppt; ++q) {
        const float m = mask[(size_t)row * n2 + tl + tpl * q];
        if (m == 1.0f) w |= 1u << q;
        else if (m != 0.0f) odd = true;
    }
    bits[i] = (uint16_t)w;
    if (odd) atomicOr(nonbinary, 1);
}


// Lane-mask tables of the wave-uniform persistent row pass (row_pipe64_kernel).  A "unit" is what one wavefront (tpl <= 64: rpw =
// 64 / tpl adjacent rows) or wpl = tpl / 64 wavefronts (one row) work on; word pipe64_word(unit, wpl, wsub, q) bit l = the mask of
// the element lane l of wavefront wsub holds in register q.
static __global__ void pack_mask64_kernel(const uint16_t* bits, unsigned long long* bits64, int n1, int tpl)
{
    const int wpl = tpl >= 64 ? tpl / 64 : 1, rpw = tpl >= 64 ? 1 : 64 / tpl;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (n1 / rpw) * wpl * 16) return;
    const int q = i & 15, wsub = (i >> 4) % wpl, unit = (i >> 4) / wpl;
    unsigned long long w = 0;
    for (int l = 0; l < 64; ++l) {
        const int row = unit * rpw + (tpl >= 64 ? 0 : l / tpl), tl = tpl >= 64 ? 64 * wsub + l : l % tpl;
        w |= (unsigned long long)((bits[(size_t)row * tpl + tl] >> q) & 1u) << l;
    }
    bits64[i] = w;
}
// cbase[word] = position in the slice's compact array of the first observed sample of that word: the compact order is unit by
// unit (rowbase of the unit's first row) and, inside a unit, word by word in table order
static __global__ void pack_cbase_kernel(const unsigned long long* bits64, const unsigned* rowbase, unsigned* cbase, int n1, int tpl)
{
    const int wpl = tpl >= 64 ? tpl / 64 : 1, rpw = tpl >= 64 ? 1 : 64 / tpl;
    const int unit = blockIdx.x * blockDim.x + threadIdx.x;
    if (unit >= n1 / rpw) return;
    unsigned run = rowbase[unit * rpw];
    for (int j = 0; j < 16 * wpl; ++j) {
        const size_t w = (size_t)unit * wpl * 16 + j;
        cbase[w] = run;
        run += (unsigned)__popcll(bits64[w]);
    }
}

// Lane-mask table of row_pipe32_kernel (rows of 1024 samples, a row pair per wavefront): word (unit u, register k) bit l =
// mask[2u + (l >> 5)][(l & 31) + 32 k], from the 16-bit words of the 64-thread layout (bits[row][tl] bit q = mask[row][tl + 64 q])
static __global__ void pack_mask32_kernel(const uint16_t* bits, unsigned long long* bits32, int n1)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (n1 / 2) * 32) return;
    const int k = i & 31, u = i >> 5;
    unsigned long long w = 0;
    for (int l = 0; l < 64; ++l) {
        const int row = 2 * u + (l >> 5), c = (l & 31) + 32 * k;
        w |= (unsigned long long)((bits[(size_t)row * 64 + (c & 63)] >> (c >> 6)) & 1u) << l;
    }
    bits32[i] = w;
}

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// the other translation units report through the same thread-local string (p3d_internal.hpp)
namespace p3d { void set_last_error(const char* msg) { g_err = msg; } }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(P3D_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct p3d_plan {
    int device = 0;
    int nil = 0, nxl = 0, max_slices = 0;
    hipStream_t stream = nullptr;
    p3d_plan* pct_plan = nullptr;      // generic twin of a tuned plan, created on first use of a percentile operator
    unsigned* pct_sel = nullptr;       // [2][max_slices][8] radix-select state
    unsigned* pct_hist = nullptr;      // [max_slices][2048]
    float* pct_frac = nullptr;         // [max_slices]
    bool generic = false;              // any-length fallback (p3d_generic.hip): row-major work buffer, unfused passes
    GenPlan gcol{}, grow{};            // factorisations of nil / nxl
    static constexpr int GEN_STAT_BLOCKS = 64;
    const LineOps* ops_col = nullptr;  // length nil (transform along iline = down the columns)
    const LineOps* ops_row = nullptr;  // length nxl (transform along xline = along the rows)
    c32 *tw_col = nullptr, *tw_row = nullptr;  // padded twiddle tables of length nil / nxl
    c32* work = nullptr;                        // column-blocked work buffer (+ 64 zero bytes behind the last slice)
    uint8_t* nzflag = nullptr;                  // [max_slices][tiles]: column-pass tile kept a coefficient (sparse skipping)
    uint16_t* nzm = nullptr;                    // [max_slices][tpl/8] the same per row-pass thread group, one bit per register
    unsigned long long* nzcount = nullptr;      // device counter: non-empty column blocks seen by nz_pack_kernel
    bool sparse_ok = false;                     // shape supports skipping emptied tiles
    double last_nonzero_fraction = -1.0;        // of the last p3d_pocs_run: kept column blocks / all (or -1: dense path)
    uint16_t* bits = nullptr;                   // packed binary trace mask [nil][tpl(nxl)]
    unsigned long long* bits64 = nullptr;       // tpl = 64, 128, 256: the same as lane masks (row_pipe64_kernel, pipe64_word)
    unsigned long long* nzl = nullptr;          // ... nzm as lane masks, per slice
    unsigned* cbase = nullptr;                  // ... observed traces before each word of bits64
    unsigned long long* mbits = nullptr;        // mixed-radix row pass (p3d_mix.hpp): packed binary mask, one word per (row, thread of the row)
    unsigned* mbase = nullptr;                  // ... and the compact-sample bases (RowArgs::mbase)
    const p3d::mix::Entry* mix_row = nullptr;   // ... the plan of the row length (nullptr: none)
    bool mix_binary = false;                    // ... the mask of the last pack_mask() was binary: mbits / mbase are valid
    // rows of 1024 samples: the one-exchange persistent row passes (row_pipe32_kernel).  use32 is fixed when the plan is created: the
    // first pass of EVERY job and of every statistics call is then that kernel's (its forward transform rounds differently from
    // line_fft<1024>; statistics and first iteration must see the same bits), whatever kernels the rest of a job takes
    bool use32 = false;
    unsigned long long* bits32 = nullptr;       // [nil / 2][32] lane masks of the row pairs (RowArgs::bits32)
    c32* tw32 = nullptr;                        // P32::build_tw
    int* flag = nullptr;                        // device int[2]: mask not binary / x non-zero at a missing trace
    unsigned* rowbase = nullptr;                // [nil+1] observed positions before each row
    void* xc = nullptr;                         // compact observed samples [nslices][nobs]
    size_t xc_cap = 0;
    double* sums = nullptr;     // [(niter+1)][nslices]
    size_t sums_cap = 0;
    double* rowsum = nullptr;   // [max_slices][nil] per-row partial sums of the current row pass
    c32* tau = nullptr;
    size_t tau_cap = 0;
    int* done = nullptr;
    float* partials = nullptr;
    int tiles = 0;
    int cus = 0;          // compute units of the device
    int pipe_wgs = 0;     // compute units handed to the persistent row pass (0: not available for this shape)
    // experiment switches read from the environment ONCE, when the plan is created (the launch paths never consult it):
    int host_sw = 0;      // P3D_SW_* bits for the flexible-length launchers
    int flex_over = 0;    // P3D_FLEX_COL_OVER (0: the launcher's default)
    bool no_colpipe = false;   // P3D_NO_COLPIPE, for the SHEARLET column pass (FFT jobs read it per job: RunSwitches)
    // staging for host-pointer entry points
    void* st_x = nullptr;
    void* st_out = nullptr;
    float* st_mask = nullptr;
    size_t st_cap = 0;
    // p3d_pocs_prime_dev: the first pass of a job run ahead of it (statistics + work buffer + compact samples of exactly this cube)
    struct Primed { bool valid = false; const void* x = nullptr; const float* mask = nullptr; int dtype = 0, nslices = 0, nonbinary = 0, violation = 0; unsigned nobs = 0;
                    bool real = false;   // the work buffer holds the half spectrum of the row pairs (float32 cube: row_real_kernel<REAL_FIRST>)
    } primed;
    double* sum0 = nullptr;     // [max_slices] sum |x_obs| of the primed cube
    int sorted_slices = 0;      // p3d_pocs_sorted_spectrum: st_x holds the sorted order keys of that many slices (0: none)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> prof_events;
    double prof_col_ms = 0, prof_row_ms = 0;
    int prof_col_n = 0, prof_row_n = 0;

    size_t slice_elems() const { return (size_t)nil * nxl; }
};

static int upload_table(p3d_plan* p, const LineOps* ops, bool for_rows, int n, c32** dst)
{
    std::vector<c32> host;
    if (is_flex(ops)) {
        flex_build_table(n, host);   // plain table exp(-2 pi i k / n), or the chirp-z tables
    } else {
        host.resize(for_rows ? (size_t)ops->row_tw_slots : (size_t)ops->col_tw_slots);
        if (for_rows) ops->build_row_tw(host.data());
        else ops->build_col_tw(host.data());
    }
    HIP_TRY(hipMalloc((void**)dst, sizeof(c32) * host.size()));
    HIP_TRY(hipMemcpy(*dst, host.data(), sizeof(c32) * host.size(), hipMemcpyHostToDevice));
    return P3D_OK;
}

extern "C" {

int p3d_abi_version(void) { return P3D_ABI_VERSION; }

const char* p3d_last_error(void) { return g_err.c_str(); }

// HIP_VERSION the library was compiled against and the version of the runtime the process actually bound (the two differ when another
// copy of libamdhip64 was mapped first, e.g. the one a PyTorch wheel brings: _ffi.py compares them)
int p3d_runtime_info(int* compiled_hip_version, int* runtime_hip_version)
{
    if (!compiled_hip_version || !runtime_hip_version) return fail(P3D_ERR_INVALID, "NULL argument");
    *compiled_hip_version = HIP_VERSION;
    HIP_TRY(hipRuntimeGetVersion(runtime_hip_version));
    return P3D_OK;
}

int p3d_device_count(int* n)
{
    if (!n) return fail(P3D_ERR_INVALID, "n is NULL");
    *n = 0;
    HIP_TRY(hipGetDeviceCount(n));
    return P3D_OK;
}

static bool generic_ok(int n) { return n >= 1 && n <= GEN_MAX_N && gen_make_plan(n).nf >= 0; }

int p3d_shape_supported(int nil, int nxl)
{
    if (find_ops(nil) != nullptr && find_ops(nxl) != nullptr) return 1;
    return (generic_ok(nil) && generic_ok(nxl)) ? 1 : 0;
}

int p3d_plan_destroy(p3d_plan* p)
{
    if (!p) return P3D_OK;
    if (p->pct_plan) p3d_plan_destroy(p->pct_plan);
    hipSetDevice(p->device);
    if (p->stream) hipStreamSynchronize(p->stream);
    void* bufs[] = {p->mbits, p->mbase, p->cbase, p->bits64, p->bits32, p->tw32, p->nzl, p->nzflag, p->nzm, p->nzcount, p->tw_col, p->tw_row, p->work, p->bits, p->flag, p->rowbase, p->xc, p->sums, p->rowsum, p->sum0, p->tau, p->pct_sel, p->pct_hist, p->pct_frac,
                    p->done,   p->partials, p->st_x, p->st_out, p->st_mask};
    for (void* b : bufs)
        if (b) hipFree(b);
    for (hipEvent_t e : p->prof_events) hipEventDestroy(e);
    if (p->ev0) hipEventDestroy(p->ev0);
    if (p->ev1) hipEventDestroy(p->ev1);
    if (p->stream) hipStreamDestroy(p->stream);
    delete p;
    return P3D_OK;
}

// force_generic: the unfused any-length pipeline even where tuned kernels exist (the percentile operators rank the whole
// spectrum, which the fused passes never materialise).  The experiment switch P3D_FORCE_GENERIC asks for the same from outside;
// it is read here, once per plan -- nothing in this library ever WRITES the process environment (plans are created and run from
// several host threads at once, and getenv racing a setenv is undefined behaviour).
static int create_plan(p3d_plan** out, int device, int nil, int nxl, int max_slices, bool force_generic)
{
    if (!out) return fail(P3D_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (nil < 1 || nxl < 1 || max_slices < 1) return fail(P3D_ERR_INVALID, "nil, nxl and max_slices must be positive");
    if (max_slices > 65535) return fail(P3D_ERR_INVALID, "max_slices > 65535: split the cube into batches");
    const LineOps* oc = find_ops(nil);
    const LineOps* orow = find_ops(nxl);
    const bool generic = !oc || !orow || force_generic;
    if (generic && !(generic_ok(nil) && generic_ok(nxl)))
        return fail(P3D_ERR_UNSUPPORTED, "slice shape %d x %d: extents up to %d are supported", nil, nxl, GEN_MAX_N);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(P3D_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));

    p3d_plan* p = new p3d_plan;
    p->device = device;
    p->nil = nil;
    p->nxl = nxl;
    p->max_slices = max_slices;
    p->generic = generic;
    p->ops_col = generic ? nullptr : oc;
    p->ops_row = generic ? nullptr : orow;
    p->host_sw = (getenv("P3D_FLEX_COL_NO_PERSIST") ? P3D_SW_FLEX_NO_PERSIST : 0) | (getenv("P3D_FLEX_NO_INPLACE") ? P3D_SW_FLEX_NO_INPLACE : 0);
    p->flex_over = getenv("P3D_FLEX_COL_OVER") ? atoi(getenv("P3D_FLEX_COL_OVER")) : 0;
    p->no_colpipe = getenv("P3D_NO_COLPIPE") != nullptr;
    if (generic) {
        p->gcol = gen_make_plan(nil);
        p->grow = gen_make_plan(nxl);
        p->tiles = p3d_plan::GEN_STAT_BLOCKS;
    } else {
        const int ct = is_flex(oc) ? flex_col_tile(nil) : oc->col_tile;
        p->tiles = (nxl + ct - 1) / ct;
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        p->cus = prop.multiProcessorCount;
        p->pipe_wgs = (orow->tpl > 0 && orow->tpl <= 256 && !getenv("P3D_NO_PIPE")) ? p->cus : 0;  // the launcher sizes the grid per variant
    }
    int rc = P3D_OK;
    auto bail = [&](int code) {
        std::string keep = g_err;
        p3d_plan_destroy(p);
        g_err = keep;
        return code;
    };
#define TRY_OR_BAIL(expr)                                                                             \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            fail(P3D_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));                         \
            return bail(P3D_ERR_HIP);                                                                 \
        }                                                                                             \
    } while (0)
    TRY_OR_BAIL(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    TRY_OR_BAIL(hipEventCreate(&p->ev0));
    TRY_OR_BAIL(hipEventCreate(&p->ev1));
    if (generic) {
        for (int which = 0; which < 2; ++which) {
            const int n = which ? nxl : nil;
            std::vector<c32> host(n);
            gen_build_twiddles(n, host.data());
            c32** dst = which ? &p->tw_row : &p->tw_col;
            TRY_OR_BAIL(hipMalloc((void**)dst, sizeof(c32) * n));
            TRY_OR_BAIL(hipMemcpy(*dst, host.data(), sizeof(c32) * n, hipMemcpyHostToDevice));
        }
        TRY_OR_BAIL(hipMalloc((void**)&p->work, sizeof(c32) * p->slice_elems() * max_slices));
        TRY_OR_BAIL(hipMalloc((void**)&p->pct_sel, sizeof(unsigned) * 16 * (size_t)max_slices));
        TRY_OR_BAIL(hipMalloc((void**)&p->pct_hist, sizeof(unsigned) * 2048 * (size_t)max_slices));
        TRY_OR_BAIL(hipMalloc((void**)&p->pct_frac, sizeof(float) * (size_t)max_slices));
    } else {
        if ((rc = upload_table(p, oc, false, nil, &p->tw_col)) != P3D_OK) return bail(rc);
        if ((rc = upload_table(p, orow, true, nxl, &p->tw_row)) != P3D_OK) return bail(rc);
        const size_t welems = wk_slice_stride(nil, nxl) * max_slices;
        TRY_OR_BAIL(hipMalloc((void**)&p->work, sizeof(c32) * welems + 64));
        TRY_OR_BAIL(hipMemset(p->work + welems, 0, 64));   // what the row pass reads for tiles the threshold emptied
        // emptied tiles can be skipped when a row-pass thread's 16 elements sit in 16 whole column blocks (nxl >= 128) and
        // 32-bit element offsets reach the zero pad
        // the tuned row pass needs a thread's 16 elements in 16 whole column blocks (nxl >= 128); the flexible one reads tile flags
        p->sparse_ok = (is_flex(orow) || orow->tpl % 8 == 0) && (double)welems + 8.0 < 4294967296.0;
        if (p->sparse_ok) {
            TRY_OR_BAIL(hipMalloc((void**)&p->nzflag, (size_t)p->tiles * max_slices));
            TRY_OR_BAIL(hipMalloc((void**)&p->nzm, sizeof(uint16_t) * (size_t)(is_flex(orow) ? 1 : orow->tpl / 8) * max_slices));
            TRY_OR_BAIL(hipMalloc((void**)&p->nzcount, sizeof(unsigned long long)));
        }
        if (orow->tpl > 0) TRY_OR_BAIL(hipMalloc((void**)&p->bits, sizeof(uint16_t) * (size_t)nil * orow->tpl));
        if (is_flex(orow) && (p->mix_row = p3d::mix::find(nxl)) != nullptr && p->mix_row->row == nullptr) p->mix_row = nullptr;   // (a columns-only plan)
        if (is_flex(orow) && p->mix_row != nullptr && !getenv("P3D_NO_MIX_BITS")) {
            TRY_OR_BAIL(hipMalloc((void**)&p->mbits, sizeof(unsigned long long) * (size_t)nil * p->mix_row->tpl));
            TRY_OR_BAIL(hipMalloc((void**)&p->mbase, sizeof(unsigned) * ((size_t)nil * p->mix_row->tpl + 1)));
        }
        if (orow->tpl >= 8 && orow->tpl <= 256 && orow->ppt == 16 && nil % (orow->tpl >= 64 ? 1 : 64 / orow->tpl) == 0 &&
            !getenv("P3D_NO_PIPE64")) {   // the wave-uniform persistent row pass: rows of 128 ... 4096 samples
            const size_t wpl = orow->tpl >= 64 ? (size_t)orow->tpl / 64 : 1;
            TRY_OR_BAIL(hipMalloc((void**)&p->bits64, sizeof(unsigned long long) * 16 * wpl * (size_t)nil));
            TRY_OR_BAIL(hipMalloc((void**)&p->cbase, sizeof(unsigned) * 16 * wpl * (size_t)nil));
            TRY_OR_BAIL(hipMalloc((void**)&p->nzl, sizeof(unsigned long long) * 16 * wpl * (size_t)max_slices));
        }
        TRY_OR_BAIL(hipMalloc((void**)&p->rowbase, sizeof(unsigned) * ((size_t)nil + 1)));
        if (orow->row_pipe32 != nullptr && p->bits64 != nullptr && p->pipe_wgs > 0 && nil % 2 == 0 && nil <= 4096 && (double)welems < 4294967296.0 &&
            !getenv("P3D_NO_PIPE32")) {
            std::vector<c32> host((size_t)orow->row_tw32_slots);
            orow->build_row_tw32(host.data());
            TRY_OR_BAIL(hipMalloc((void**)&p->tw32, sizeof(c32) * host.size()));
            TRY_OR_BAIL(hipMemcpy(p->tw32, host.data(), sizeof(c32) * host.size(), hipMemcpyHostToDevice));
            TRY_OR_BAIL(hipMalloc((void**)&p->bits32, sizeof(unsigned long long) * 32 * (size_t)(nil / 2)));
            p->use32 = true;
        }
    }
    TRY_OR_BAIL(hipMalloc((void**)&p->flag, 2 * sizeof(int)));
    TRY_OR_BAIL(hipMalloc((void**)&p->rowsum, sizeof(double) * (size_t)nil * max_slices));
    TRY_OR_BAIL(hipMalloc((void**)&p->sum0, sizeof(double) * (size_t)max_slices));
    TRY_OR_BAIL(hipMalloc((void**)&p->done, sizeof(int) * max_slices));
    // (the paired shearlet statistics pass writes one record per 8 columns whatever the column tile of the length)
    TRY_OR_BAIL(hipMalloc((void**)&p->partials, sizeof(float) * STATS_PARTIAL * (size_t)std::max(p->tiles, (nxl + 7) / 8) * max_slices));
#undef TRY_OR_BAIL
    *out = p;
    return P3D_OK;
}

int p3d_plan_create(p3d_plan** out, int device, int nil, int nxl, int max_slices)
{
    return create_plan(out, device, nil, nxl, max_slices, getenv("P3D_FORCE_GENERIC") != nullptr);
}

int p3d_malloc(p3d_plan* p, void** dptr, size_t bytes)
{
    if (!p || !dptr) return fail(P3D_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMalloc(dptr, bytes));
    return P3D_OK;
}

int p3d_free(p3d_plan* p, void* dptr)
{
    // p3d_malloc is a bare hipMalloc: the buffer does not die with its plan, so it can (and must) be freed after the plan is gone
    // too -- `p` may be NULL then (hipFree finds the owning device from the pointer)
    if (p) HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipFree(dptr));
    return P3D_OK;
}

int p3d_dev_malloc(int device, void** dptr, size_t bytes)
{
    if (!dptr) return fail(P3D_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc(dptr, bytes));
    return P3D_OK;
}

int p3d_dev_free(void* dptr)
{
    HIP_TRY(hipFree(dptr));
    return P3D_OK;
}

int p3d_dev_memcpy(int device, void* dst, const void* src, size_t bytes, int kind)
{
    if (kind < 0 || kind > 2) return fail(P3D_ERR_INVALID, "kind must be 0 (h2d), 1 (d2h) or 2 (d2d)");
    if (!bytes) return P3D_OK;
    if (!dst || !src) return fail(P3D_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(dst, src, bytes, kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice));
    if (kind == 2) HIP_TRY(hipDeviceSynchronize());   // (device-to-device copies return before they have run)
    return P3D_OK;
}

int p3d_dev_memset(int device, void* dptr, int value, size_t bytes)
{
    if (!bytes) return P3D_OK;
    if (!dptr) return fail(P3D_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemset(dptr, value, bytes));
    HIP_TRY(hipDeviceSynchronize());
    return P3D_OK;
}

int p3d_dev_synchronize(int device)
{
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    return P3D_OK;
}

int p3d_dev_mem_info(int device, size_t* free_bytes, size_t* total_bytes)
{
    if (!free_bytes || !total_bytes) return fail(P3D_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
    return P3D_OK;
}

int p3d_host_alloc(void** hptr, size_t bytes)
{
    if (!hptr) return fail(P3D_ERR_INVALID, "NULL argument");
    HIP_TRY(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return P3D_OK;
}

int p3d_host_free(void* hptr)
{
    HIP_TRY(hipHostFree(hptr));
    return P3D_OK;
}

// Both copies run on the PLAN's stream and return when they have completed.  hipMemcpy would run them on the null stream, where the
// copies of all plans of a process queue up behind each other: four chunk workers then share ONE direction of the link at a time
// (measured: uploads + downloads of a 4-GiB cube at 56 GB/s in total; on separate streams, from page-locked or registered memory,
// 96 GB/s -- tools/pcie_probe2.py, profiles/r04_pcie_probe.txt).
int p3d_memcpy_h2d(p3d_plan* p, void* dst, const void* src, size_t bytes)
{
    if (!p) return fail(P3D_ERR_INVALID, "NULL plan");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return P3D_OK;
}

int p3d_memcpy_d2h(p3d_plan* p, void* dst, const void* src, size_t bytes)
{
    if (!p) return fail(P3D_ERR_INVALID, "NULL plan");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return P3D_OK;
}

// Page-lock a caller's host array in place for the duration of a job (hipHostRegister): copies to / from it are then plain DMA
// transfers that run concurrently in both directions.  8.8 ms for a touched 4-GiB array; a FRESH allocation should be touched first
// (the driver faults its pages one by one: 176 ms, against 24 ms for eight threads writing one byte per page).
// P3D_ERR_UNSUPPORTED: the runtime refuses the range (already registered, read-only mapping ...): the caller goes on without.
int p3d_host_register(void* hptr, size_t bytes)
{
    if (!hptr || bytes == 0) return fail(P3D_ERR_INVALID, "NULL argument");
    const hipError_t e = hipHostRegister(hptr, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(P3D_ERR_UNSUPPORTED, "hipHostRegister refused %zu bytes at %p: %s", bytes, hptr, hipGetErrorString(e));
    }
    return P3D_OK;
}

int p3d_host_unregister(void* hptr)
{
    if (!hptr) return fail(P3D_ERR_INVALID, "NULL argument");
    HIP_TRY(hipHostUnregister(hptr));
    return P3D_OK;
}

}  // extern "C"

// ---- internal helpers ---------------------------------------------------------------------------
// Diagnostic switches of one job: every one selects a slower, equivalent path.  Read once per call of
// p3d_pocs_run_dev, on the calling thread, and handed down as plain values -- the launch paths never consult the environment.
struct RunSwitches { bool no_pct_fused; bool no_mask_bits, no_compact, no_real, no_sparse, real_2048, no_resident, no_tstore, force_colpipe, no_colpipe; };
static RunSwitches read_switches()
{
    RunSwitches s;
    s.no_pct_fused = getenv("P3D_NO_PCT_FUSED") != nullptr;
    s.no_mask_bits = getenv("P3D_NO_MASK_BITS") != nullptr;
    s.no_compact = getenv("P3D_NO_COMPACT") != nullptr;
    s.no_real = getenv("P3D_NO_REAL") != nullptr;
    s.no_sparse = getenv("P3D_NO_SPARSE") != nullptr;
    s.real_2048 = getenv("P3D_NO_REAL_2048") == nullptr;   // rows of 2048 samples in pairs: on since the kernel got the registers its occupancy allows (round 3)
    s.no_resident = getenv("P3D_NO_RESIDENT") != nullptr;
    s.no_tstore = getenv("P3D_NO_TSTORE") != nullptr;
    s.force_colpipe = getenv("P3D_FORCE_COLPIPE") != nullptr;
    s.no_colpipe = getenv("P3D_NO_COLPIPE") != nullptr;
    return s;
}

static int check_batch(p3d_plan* p, int nslices)
{
    if (!p) return fail(P3D_ERR_INVALID, "NULL plan");
    p->primed.valid = false;   // whoever comes through here is about to use the work buffer (p3d_pocs_run_dev reads the state first)
    p->sorted_slices = 0;      // ... and possibly the staging buffers (p3d_pocs_data_driven_pick reads the state first)
    if (nslices < 1 || nslices > p->max_slices)
        return fail(P3D_ERR_INVALID, "nslices = %d outside 1..max_slices (%d)", nslices, p->max_slices);
    return P3D_OK;
}

static RowArgs row_args(p3d_plan* p, int nslices)
{
    RowArgs r{};
    r.tw = p->tw_row;
    r.tw32 = p->tw32;
    r.n1 = p->nil;
    r.nslices = nslices;
    r.alpha = 1.0f;
    r.scale = (float)(1.0 / ((double)p->nil * (double)p->nxl));
    r.len = p->nxl;
    r.host_sw = p->host_sw;
    return r;
}

static ColArgs col_args(p3d_plan* p, int nslices)
{
    ColArgs c{};
    c.tw = p->tw_col;
    c.n2 = p->nxl;
    c.nslices = nslices;
    c.len = p->nil;
    c.cus = p->cus;
    c.host_sw = p->host_sw;
    c.flex_over = p->flex_over;
    return c;
}

static int ensure_staging(p3d_plan* p, size_t bytes_per_cube)
{
    if (p->st_cap >= bytes_per_cube && p->st_x) return P3D_OK;
    if (p->st_x) hipFree(p->st_x);
    if (p->st_out) hipFree(p->st_out);
    p->st_x = p->st_out = nullptr;
    p->st_cap = 0;
    HIP_TRY(hipMalloc(&p->st_x, bytes_per_cube));
    HIP_TRY(hipMalloc(&p->st_out, bytes_per_cube));
    if (!p->st_mask) HIP_TRY(hipMalloc((void**)&p->st_mask, sizeof(float) * p->slice_elems()));
    p->st_cap = bytes_per_cube;
    return P3D_OK;
}


// ---- any-length fallback (p3d_generic.hip) -------------------------------------------------------------------
static int gen_fft2(p3d_plan* p, const c32* in, c32* out, int nslices, int inverse, const int* done)
{
    const float scale = (float)(1.0 / ((double)p->nil * (double)p->nxl));
    if (!inverse) {
        HIP_TRY(gen_launch_line_fft(in, out, p->tw_row, p->grow, FWD, 1.0f, nslices, p->nil, p->nxl, true, done, p->stream));
        HIP_TRY(gen_launch_line_fft(out, out, p->tw_col, p->gcol, FWD, 1.0f, nslices, p->nil, p->nxl, false, done, p->stream));
    } else {
        HIP_TRY(gen_launch_line_fft(in, out, p->tw_col, p->gcol, INV, 1.0f, nslices, p->nil, p->nxl, false, done, p->stream));
        HIP_TRY(gen_launch_line_fft(out, out, p->tw_row, p->grow, INV, scale, nslices, p->nil, p->nxl, true, done, p->stream));
    }
    return P3D_OK;
}

static int reduce_partials(p3d_plan* p, int nslices, double* stats, int tiles = 0)
{
    if (tiles <= 0) tiles = p->tiles;
    std::vector<float> part((size_t)STATS_PARTIAL * tiles * nslices);
    HIP_TRY(hipMemcpyAsync(part.data(), p->partials, sizeof(float) * part.size(), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (int s = 0; s < nslices; ++s) {
        double lr = -INFINITY, li = -INFINITY, mx = 0.0, mn = INFINITY, sq = 0.0;
        for (int t = 0; t < tiles; ++t) {
            const float* q = &part[((size_t)s * tiles + t) * STATS_PARTIAL];
            if (q[0] > lr || (q[0] == lr && q[1] > li)) { lr = q[0]; li = q[1]; }
            if (q[2] > mx) mx = q[2];
            if (q[3] < mn) mn = q[3];
            sq += q[4];
        }
        double* o = stats + (size_t)s * P3D_STATS_PER_SLICE;
        o[0] = lr; o[1] = li; o[2] = mx; o[3] = mn; o[4] = sq; o[5] = 0.0;
    }
    return P3D_OK;
}

static hipError_t first_row_pass(p3d_plan* p, const RowArgs& r);

static int fft2_enqueue(p3d_plan* p, const void* in, void* out, int nslices, int inverse)
{
    int rc = check_batch(p, nslices);
    if (rc) return rc;
    if (!in || !out) return fail(P3D_ERR_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(p->device));
    if (p->generic) {
        return gen_fft2(p, (const c32*)in, (c32*)out, nslices, inverse, nullptr);
    }
    if (!inverse) {
        RowArgs r = row_args(p, nslices);
        r.x = in;
        r.work = p->work;
        r.dtype = P3D_C64;
        HIP_TRY(first_row_pass(p, r));   // (every forward transform of a plan takes the same row pass: the same bits as the statistics and the loop)
        ColArgs c = col_args(p, nslices);
        c.in = p->work;
        c.out = (c32*)out;
        c.out_std = 1;
        HIP_TRY(p->ops_col->col(COL_FWD, c, p->stream));
    } else {
        ColArgs c = col_args(p, nslices);
        c.in = (const c32*)in;
        c.in_std = 1;
        c.out = p->work;
        HIP_TRY(p->ops_col->col(COL_INV, c, p->stream));
        RowArgs r = row_args(p, nslices);
        r.work = p->work;
        r.out = out;
        r.dtype = P3D_C64;
        r.plain = 1;
        HIP_TRY(p->ops_row->row(ROW_LAST, r, p->stream));
    }
    return P3D_OK;
}

namespace p3d {
// ---- the three fused passes of one SHEARLET iteration (p3d_shearlet.hip); power-of-two plans only ----------------------------
bool shearlet_fused_supported(p3d_plan* plan) { return plan && !plan->generic && plan->ops_row->tpl > 0 && plan->ops_col->tpl > 0; }

static ShearArgs shear_args(const float* psi, const c32* tau, int nsh, int niter, int iter, int op, int real_only, const unsigned* sup, int sup_words, bool half)
{
    ShearArgs a{};
    a.psi = psi; a.tau = tau; a.nsh = nsh; a.niter = niter; a.iter = iter; a.op = op; a.real_only = real_only;
    a.sup = sup; a.sup_words = sup_words; a.half = half ? 1 : 0;
    return a;
}
bool shearlet_pair_supported(p3d_plan* p)
{
    return shearlet_fused_supported(p) && p->ops_col->col_shear_pair != nullptr && p->ops_col->tpl >= 32 && p->ops_col->tpl <= 256 && p->nxl % 8 == 0 &&
           (double)wk_slice_stride(p->nil, p->nxl) < 4294967296.0 / 8.0;
}

int shearlet_spread_inv(p3d_plan* p, const c32* F, const float* psi, int nb, int nsh, const unsigned* sup, int sup_words, bool pair)
{
    int rc = check_batch(p, nb * nsh);
    if (rc) return rc;
    RowArgs r = row_args(p, nb);   // one workgroup row-group per slice; the shearlets are looped over inside
    r.x = F;
    r.work = p->work;
    r.sh = shear_args(psi, nullptr, nsh, 0, 0, 0, 0, sup, sup_words, pair);
    HIP_TRY(p->ops_row->row(ROW_SPREAD_INV, r, p->stream));
    return P3D_OK;
}

int shearlet_col_shrink(p3d_plan* p, const c32* tau, int nb, int nsh, int niter, int iter, int op, int real_only, const unsigned* sup, int sup_words, bool pair)
{
    int rc = check_batch(p, nb * nsh);
    if (rc) return rc;
    ColArgs c = col_args(p, nb * nsh);
    c.in = p->work;
    c.out = p->work;
    c.sh = shear_args(nullptr, tau, nsh, niter, iter, op, real_only, sup, sup_words, pair);
    // columns of 2048 points and more: one 8-column tile per CU, so the persistent pass (next tile requested while this one is
    // transformed) has something to give (see p3d_pocs_run_dev); P3D_NO_COLPIPE=1 (read when the plan is created) switches it off
    hipError_t ce = hipErrorNotSupported;
    if (pair) {   // two columns per transform on Hermitian work slices: the other two passes ran / will run on half the rows, no way back
        if (!real_only || !shearlet_pair_supported(p)) return fail(P3D_ERR_INVALID, "the paired column pass does not apply to this plan");
        HIP_TRY(p->ops_col->col_shear_pair(c, p->stream));
        return P3D_OK;
    }
    if (!p->no_colpipe && p->cus > 0 && p->ops_col->col_pipe != nullptr && p->ops_col->n >= 2048) ce = p->ops_col->col_pipe(c, p->cus, p->stream);
    if (ce == hipErrorNotSupported) ce = p->ops_col->col(COL_SHRINK, c, p->stream);
    HIP_TRY(ce);
    return P3D_OK;
}

int shearlet_col_stats_pair(p3d_plan* p, int nb, int nsh, const unsigned* sup, int sup_words, float* host_stats)
{
    int rc = check_batch(p, nb * nsh);
    if (rc) return rc;
    if (!shearlet_pair_supported(p) || !host_stats) return fail(P3D_ERR_INVALID, "the paired column pass does not apply to this plan");
    ColArgs c = col_args(p, nb * nsh);
    c.in = p->work;
    c.out = p->work;
    c.partials = p->partials;
    c.sh = shear_args(nullptr, nullptr, nsh, 0, 0, 0, 1, sup, sup_words, true);
    HIP_TRY(p->ops_col->col_shear_pair(c, p->stream));
    const int tiles = p->nxl / 8;
    std::vector<float> part((size_t)STATS_PARTIAL * tiles * nb * nsh);
    HIP_TRY(hipMemcpyAsync(part.data(), p->partials, sizeof(float) * part.size(), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (int bs = 0; bs < nb * nsh; ++bs) {
        float smax = -INFINITY, mx = 0.f, mn = INFINITY;
        double sq = 0.0;
        for (int t = 0; t < tiles; ++t) {
            const float* q = &part[((size_t)bs * tiles + t) * STATS_PARTIAL];
            smax = std::max(smax, q[0]); mx = std::max(mx, q[2]); mn = std::min(mn, q[3]); sq += q[4];
        }
        float* o = host_stats + (size_t)bs * 5;
        o[0] = smax; o[1] = 0.f; o[2] = std::sqrt(mx); o[3] = std::sqrt(mn); o[4] = (float)sq;
    }
    return P3D_OK;
}

int shearlet_gather_fwd(p3d_plan* p, const float* psi, c32* out, int nb, int nsh, const unsigned* sup, int sup_words, bool pair)
{
    int rc = check_batch(p, nb * nsh);
    if (rc) return rc;
    RowArgs r = row_args(p, nb);
    r.work = p->work;
    r.out = out;
    r.sh = shear_args(psi, nullptr, nsh, 0, 0, 0, 0, sup, sup_words, pair);
    HIP_TRY(p->ops_row->row(ROW_GATHER_FWD, r, p->stream));
    return P3D_OK;
}

hipStream_t plan_stream(p3d_plan* plan) { return plan->stream; }
int fft2_async(p3d_plan* plan, const c32* in, c32* out, int nslices, int inverse) { return fft2_enqueue(plan, in, out, nslices, inverse); }
}  // namespace p3d

extern "C" {

int p3d_fft2_c64_dev(p3d_plan* p, const void* in, void* out, int nslices, int inverse)
{
    int rc = fft2_enqueue(p, in, out, nslices, inverse);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));
    return P3D_OK;
}

int p3d_fft2_c64(p3d_plan* p, const void* in, void* out, int nslices, int inverse)
{
    int rc = check_batch(p, nslices);
    if (rc) return rc;
    if (!in || !out) return fail(P3D_ERR_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(p->device));
    const size_t bytes = sizeof(c32) * p->slice_elems() * nslices;
    if ((rc = ensure_staging(p, sizeof(c32) * p->slice_elems() * p->max_slices))) return rc;
    HIP_TRY(hipMemcpy(p->st_x, in, bytes, hipMemcpyHostToDevice));
    if ((rc = p3d_fft2_c64_dev(p, p->st_x, p->st_out, nslices, inverse))) return rc;
    HIP_TRY(hipMemcpy(out, p->st_out, bytes, hipMemcpyDeviceToHost));
    return P3D_OK;
}

// 'data-driven' schedule (POCS.py:356-362) without downloading the spectrum: see include/p3d.h and p3d_select.hip
int p3d_pocs_sorted_spectrum(p3d_plan* p, const void* x, int nslices, float* peaks)
{
    int rc = check_batch(p, nslices);
    if (rc) return rc;
    if (!x || !peaks) return fail(P3D_ERR_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(p->device));
    const size_t per = p->slice_elems(), bytes = sizeof(c32) * per * nslices;
    if (per > 0xffffffffull) return fail(P3D_ERR_UNSUPPORTED, "%zu samples per slice: the segmented sort indexes a slice with 32 bits", per);
    if ((rc = ensure_staging(p, sizeof(c32) * per * p->max_slices))) return rc;
    HIP_TRY(hipMemcpyAsync(p->st_x, x, bytes, hipMemcpyDefault, p->stream));   // (x may be a device pointer: ordered with the plan's stream)
    if ((rc = fft2_enqueue(p, p->st_x, p->st_out, nslices, 0))) return rc;
    float* dpeaks = nullptr;
    HIP_TRY(hipMalloc((void**)&dpeaks, sizeof(float) * 2 * nslices));
    const hipError_t e = lex_sort_desc((c32*)p->st_out, p->st_x, per, nslices, dpeaks, p->stream);   // keys in st_out, sorted in st_x
    hipError_t e2 = hipSuccess;
    if (e == hipSuccess) e2 = hipMemcpy(peaks, dpeaks, sizeof(float) * 2 * nslices, hipMemcpyDeviceToHost);
    hipFree(dpeaks);
    HIP_TRY(e);
    HIP_TRY(e2);
    p->sorted_slices = nslices;
    return P3D_OK;
}

int p3d_pocs_data_driven_pick(p3d_plan* p, int nslices, int niter, const float* bounds, float* tau, int64_t* count)
{
    if (!p) return fail(P3D_ERR_INVALID, "NULL plan");
    if (!bounds || !tau || !count || niter < 1) return fail(P3D_ERR_INVALID, "NULL buffer or niter < 1");
    if (p->sorted_slices != nslices || nslices < 1)
        return fail(P3D_ERR_INVALID, "p3d_pocs_sorted_spectrum has not just run on %d slices of this plan", nslices);
    HIP_TRY(hipSetDevice(p->device));
    float *dbounds = nullptr, *dtau = nullptr;
    long long* dcount = nullptr;
    hipError_t e = hipMalloc((void**)&dbounds, sizeof(float) * 4 * nslices);
    if (e == hipSuccess) e = hipMalloc((void**)&dtau, sizeof(float) * 2 * (size_t)niter * nslices);
    if (e == hipSuccess) e = hipMalloc((void**)&dcount, sizeof(long long) * nslices);
    if (e == hipSuccess) e = hipMemsetAsync(dtau, 0, sizeof(float) * 2 * (size_t)niter * nslices, p->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dbounds, bounds, sizeof(float) * 4 * nslices, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) e = data_driven_pick(p->st_x, p->slice_elems(), nslices, niter, dbounds, dtau, dcount, p->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(tau, dtau, sizeof(float) * 2 * (size_t)niter * nslices, hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(count, dcount, sizeof(long long) * nslices, hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    if (dbounds) hipFree(dbounds);
    if (dtau) hipFree(dtau);
    if (dcount) hipFree(dcount);
    HIP_TRY(e);
    return P3D_OK;
}

int p3d_fft2_shrink_c64(p3d_plan* p, const void* in, const double* tau, int op, void* out, int nslices)
{
    int rc = check_batch(p, nslices);
    if (rc) return rc;
    if (!in || !out || !tau) return fail(P3D_ERR_INVALID, "NULL buffer");
    if (op < P3D_OP_HARD || op > P3D_OP_GARROTE) return fail(P3D_ERR_UNSUPPORTED, "thresh_op %d is not implemented", op);
    HIP_TRY(hipSetDevice(p->device));
    const size_t bytes = sizeof(c32) * p->slice_elems() * nslices;
    if ((rc = ensure_staging(p, sizeof(c32) * p->slice_elems() * p->max_slices))) return rc;
    if (p->tau_cap < (size_t)nslices) {
        if (p->tau) hipFree(p->tau);
        p->tau = nullptr;
        p->tau_cap = 0;
        HIP_TRY(hipMalloc((void**)&p->tau, sizeof(c32) * nslices));
        p->tau_cap = nslices;
    }
    std::vector<c32> tau_f(nslices);
    for (int s = 0; s < nslices; ++s) tau_f[s] = tau_for_device(tau[2 * s], tau[2 * s + 1], op == P3D_OP_HARD);
    HIP_TRY(hipMemcpy(p->tau, tau_f.data(), sizeof(c32) * nslices, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->st_x, in, bytes, hipMemcpyHostToDevice));
    if (p->generic) {
        if ((rc = gen_fft2(p, (const c32*)p->st_x, (c32*)p->st_out, nslices, 0, nullptr))) return rc;
        HIP_TRY(gen_launch_shrink((c32*)p->st_out, p->tau, 1, 0, op, nslices, p->slice_elems(), nullptr, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        HIP_TRY(hipMemcpy(out, p->st_out, bytes, hipMemcpyDeviceToHost));
        return P3D_OK;
    }
    RowArgs r = row_args(p, nslices);
    r.x = p->st_x;
    r.work = p->work;
    r.dtype = P3D_C64;
    HIP_TRY(first_row_pass(p, r));
    ColArgs c = col_args(p, nslices);
    c.in = p->work;
    c.out = (c32*)p->st_out;
    c.out_std = 1;
    c.tau = p->tau;
    c.niter = 1;
    c.iter = 0;
    c.op = op;
    HIP_TRY(p->ops_col->col(COL_FWD, c, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy(out, p->st_out, bytes, hipMemcpyDeviceToHost));
    return P3D_OK;
}

// first / last row pass of a job on the complex path: the wave-uniform persistent kernel where the plan has its tables
// (rows of 128 ... 4096 samples, P3D_NO_PIPE64 unset), the one-launch-per-pass kernels otherwise
static hipError_t first_row_pass(p3d_plan* p, const RowArgs& r)
{
    if (p->use32) {   // rows of 1024 samples: always this first pass where the plan has it (see p3d_plan::use32)
        const hipError_t e = p->ops_row->row_pipe32(PIPE_FIRST, r, p->pipe_wgs, p->stream);
        if (e != hipErrorNotSupported) return e;   // (not supported: APOCS on a non-binary mask -- the input mix needs the mask bits)
    }
    if (p->pipe_wgs > 0 && p->bits64 != nullptr && p->ops_row->row_pipe64 != nullptr) {
        const hipError_t e = p->ops_row->row_pipe64(PIPE_FIRST, r, p->pipe_wgs, p->stream);
        if (e != hipErrorNotSupported) return e;
    }
    return p->ops_row->row(ROW_FIRST, r, p->stream);
}

// packed forms of a binary trace mask (16-bit words per thread and row, 64-bit lane masks, compact bases); nonbinary: the mask holds
// other values than 0 / 1 (or the row pass is a flexible one, which reads the float weights); nobs: observed positions per slice
static int pack_mask(p3d_plan* p, const float* mask, int* nonbinary, unsigned* nobs)
{
    const bool flex_rows = is_flex(p->ops_row);
    HIP_TRY(hipMemsetAsync(p->flag, 0, 2 * sizeof(int), p->stream));
    *nonbinary = flex_rows ? 1 : 0;
    *nobs = 0;
    p->mix_binary = false;
    if (flex_rows && p->mbits) {
        // rows on the mixed-radix register engine: their own packed words; the float weights stay with every other flexible row pass
        // (`nonbinary` keeps saying so: the callers hand RowArgs::mask on), p->mix_binary says whether the words may be used beside them
        int odd = 0;
        HIP_TRY(p3d::mix::pack_mask(p->mix_row, mask, p->nil, p->mbits, p->mbase, p->flag, p->stream));
        HIP_TRY(hipMemcpyAsync(&odd, p->flag, sizeof(int), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipMemcpyAsync(nobs, p->mbase + (size_t)p->nil * p->mix_row->tpl, sizeof(unsigned), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        p->mix_binary = odd == 0;
        HIP_TRY(hipMemsetAsync(p->flag, 0, 2 * sizeof(int), p->stream));
        return P3D_OK;
    }
    if (flex_rows) return P3D_OK;
    const int words = p->nil * p->ops_row->tpl;
    pack_mask_kernel<<<(words + 255) / 256, 256, 0, p->stream>>>(mask, p->bits, p->flag, p->nil, p->nxl, p->ops_row->tpl, p->ops_row->ppt);
    if (p->nil <= 4096) rowbase_kernel<<<1, 1024, 0, p->stream>>>(p->bits, p->rowbase, p->nil, p->ops_row->tpl);
    if (p->bits64 && p->nil <= 4096) {
        const int tpl = p->ops_row->tpl, wpl = tpl >= 64 ? tpl / 64 : 1;
        pack_mask64_kernel<<<(p->nil * wpl * 16 + 255) / 256, 256, 0, p->stream>>>(p->bits, p->bits64, p->nil, tpl);
        pack_cbase_kernel<<<(p->nil + 255) / 256, 256, 0, p->stream>>>(p->bits64, p->rowbase, p->cbase, p->nil, tpl);
    }
    if (p->use32) pack_mask32_kernel<<<((p->nil / 2) * 32 + 255) / 256, 256, 0, p->stream>>>(p->bits, p->bits32, p->nil);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(nonbinary, p->flag, sizeof(int), hipMemcpyDeviceToHost, p->stream));
    if (p->nil <= 4096) HIP_TRY(hipMemcpyAsync(nobs, p->rowbase + p->nil, sizeof(unsigned), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return P3D_OK;
}

// compact observed samples are worth their bookkeeping when the persistent row pass exists and less than 3/4 of the traces are there
static bool want_compact(p3d_plan* p, int nonbinary, unsigned nobs, int niter, const RunSwitches& sw)
{
    return !nonbinary && p->pipe_wgs > 0 && p->nil <= 4096 && niter > 1 && !sw.no_compact && nobs > 0 &&
           (double)nobs < 0.75 * (double)p->slice_elems();
}

static int ensure_xc(p3d_plan* p, int nslices, unsigned nobs, int dtype)
{
    const size_t need = (size_t)nslices * nobs * (dtype == P3D_C64 ? sizeof(c32) : sizeof(float));
    if (p->xc_cap < need) {
        if (p->xc) hipFree(p->xc);
        p->xc = nullptr;
        p->xc_cap = 0;
        HIP_TRY(hipMalloc(&p->xc, need));
        p->xc_cap = need;
    }
    return P3D_OK;
}

int p3d_pocs_prime_dev(p3d_plan* p, const void* x, int dtype, const float* mask, int nslices, double* stats)
{
    int rc = check_batch(p, nslices);
    if (rc) return rc;
    if (!x || !mask || !stats) return fail(P3D_ERR_INVALID, "NULL buffer");
    if (dtype != P3D_C64 && dtype != P3D_F32) return fail(P3D_ERR_INVALID, "unknown dtype %d", dtype);
    HIP_TRY(hipSetDevice(p->device));
    if (p->generic || is_flex(p->ops_row) || is_flex(p->ops_col)) return p3d_pocs_stats_dev(p, x, dtype, nslices, stats);   // nothing to run ahead there
    const RunSwitches sw = read_switches();
    int nonbinary = 0;
    unsigned nobs = 0;
    if ((rc = pack_mask(p, mask, &nonbinary, &nobs))) return rc;
    if (sw.no_mask_bits) nonbinary = 1;
    const bool compact = want_compact(p, nonbinary, nobs, 2, sw);
    if (compact && (rc = ensure_xc(p, nslices, nobs, dtype))) return rc;
    RowArgs r = row_args(p, nslices);
    r.x = x;
    r.mask = nonbinary ? mask : nullptr;
    r.bits = nonbinary ? nullptr : p->bits;
    r.bits64 = nonbinary ? nullptr : p->bits64;
    r.cbase = nonbinary ? nullptr : p->cbase;
    r.bits32 = (nonbinary || !p->use32) ? nullptr : p->bits32;
    r.xc = compact ? p->xc : nullptr;
    r.rowbase = p->rowbase;
    r.nobs = nobs;
    r.violation = p->flag + 1;
    r.work = p->work;
    r.sums = p->rowsum;
    r.dtype = dtype;
    r.real_2048 = sw.real_2048 ? 1 : 0;
    HIP_TRY(hipMemsetAsync(p->rowsum, 0, sizeof(double) * (size_t)p->nil * nslices, p->stream));
    // float32 cubes: the row pairs of the real path (half the transforms, half-spectrum work buffer) and the statistics of the whole
    // Hermitian spectrum from its stored half (ColArgs::herm_n2).  A run that takes the real path too (hard operator, POCS / FPOCS)
    // finds its first pass done; any other run starts over with the complex first pass.
    bool real = dtype == P3D_F32 && !sw.no_real && compact && r.bits64 && r.cbase && p->pipe_wgs > 0 && p->ops_row->row_real != nullptr;
    if (real) {
        const hipError_t re = p->ops_row->row_real(REAL_FIRST, r, p->pipe_wgs, p->stream);
        if (re == hipErrorNotSupported) real = false;
        else HIP_TRY(re);
    }
    int violation = 0;
    if (real) {   // the row pairs live on the compact samples: a cube with energy at unobserved positions takes the complex path
        HIP_TRY(hipMemcpyAsync(&violation, p->flag + 1, sizeof(int), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        if (violation) {
            real = false;
            HIP_TRY(hipMemsetAsync(p->rowsum, 0, sizeof(double) * (size_t)p->nil * nslices, p->stream));
        }
    }
    if (!real) HIP_TRY(first_row_pass(p, r));
    reduce_rows_kernel<<<nslices, 256, 0, p->stream>>>(p->rowsum, p->sum0, p->nil);
    ColArgs c = col_args(p, nslices);
    c.in = p->work;
    c.partials = p->partials;
    int tiles = p->tiles;
    if (real) {
        c.n2 = p->nxl / 2 + 1;
        c.herm_n2 = p->nxl;
        tiles = (c.n2 + p->ops_col->col_tile - 1) / p->ops_col->col_tile;
    }
    HIP_TRY(p->ops_col->col(COL_STATS, c, p->stream));
    if (compact && !real) HIP_TRY(hipMemcpyAsync(&violation, p->flag + 1, sizeof(int), hipMemcpyDeviceToHost, p->stream));
    if ((rc = reduce_partials(p, nslices, stats, tiles))) return rc;   // (synchronises the stream)
    p->primed.valid = true;
    p->primed.x = x; p->primed.mask = mask; p->primed.dtype = dtype; p->primed.nslices = nslices;
    p->primed.nonbinary = nonbinary; p->primed.nobs = nobs; p->primed.violation = violation;
    p->primed.real = real;
    return P3D_OK;
}

int p3d_pocs_stats_dev(p3d_plan* p, const void* x, int dtype, int nslices, double* stats)
{
    int rc = check_batch(p, nslices);
    if (rc) return rc;
    if (!x || !stats) return fail(P3D_ERR_INVALID, "NULL buffer");
    if (dtype != P3D_C64 && dtype != P3D_F32) return fail(P3D_ERR_INVALID, "unknown dtype %d", dtype);
    HIP_TRY(hipSetDevice(p->device));
    if (p->generic) {
        HIP_TRY(hipMemsetAsync(p->rowsum, 0, sizeof(double) * nslices, p->stream));
        HIP_TRY(gen_launch_update(p->work, x, dtype, nullptr, nullptr, p->rowsum, 0, 0, 0, 1.0f, nslices, p->slice_elems(), nullptr, 0,
                                  p->stream));
        if ((rc = gen_fft2(p, p->work, p->work, nslices, 0, nullptr))) return rc;
        HIP_TRY(gen_launch_stats(p->work, p->partials, nslices, p->slice_elems(), p->tiles, p->stream));
        return reduce_partials(p, nslices, stats);
    }
    RowArgs r = row_args(p, nslices);
    r.x = x;
    r.work = p->work;
    r.dtype = dtype;
    HIP_TRY(first_row_pass(p, r));
    ColArgs c = col_args(p, nslices);
    c.in = p->work;
    c.partials = p->partials;
    HIP_TRY(p->ops_col->col(COL_STATS, c, p->stream));
    return reduce_partials(p, nslices, stats);
}

int p3d_pocs_stats(p3d_plan* p, const void* x, int dtype, int nslices, double* stats)
{
    int rc = check_batch(p, nslices);
    if (rc) return rc;
    if (!x || !stats) return fail(P3D_ERR_INVALID, "NULL buffer");
    if (dtype != P3D_C64 && dtype != P3D_F32) return fail(P3D_ERR_INVALID, "unknown dtype %d", dtype);
    HIP_TRY(hipSetDevice(p->device));
    if ((rc = ensure_staging(p, sizeof(c32) * p->slice_elems() * p->max_slices))) return rc;
    const size_t esz = dtype == P3D_C64 ? sizeof(c32) : sizeof(float);
    HIP_TRY(hipMemcpy(p->st_x, x, esz * p->slice_elems() * nslices, hipMemcpyHostToDevice));
    return p3d_pocs_stats_dev(p, p->st_x, dtype, nslices, stats);
}

int p3d_pocs_run_dev(p3d_plan* p, const void* x, int dtype, const float* mask, const double* tau, const uint8_t* active,
                     const p3d_pocs_params* prm, void* out, int nslices, int32_t* niter_done, double* sums,
                     double* elapsed_ms)
{
    // P3D_FLAG_PRIMED: the caller says p3d_pocs_prime_dev has just run on exactly this cube and mask -- believed only if the plan
    // agrees (same pointers, type and batch, nothing else on the plan in between)
    const bool primed_in = p && prm && (prm->flags & P3D_FLAG_PRIMED) && p->primed.valid && p->primed.x == x && p->primed.mask == mask &&
                           p->primed.dtype == dtype && p->primed.nslices == nslices;
    const p3d_plan::Primed primed_state = p ? p->primed : p3d_plan::Primed{};
    int rc = check_batch(p, nslices);
    if (rc) return rc;
    if (!x || !mask || !tau || !prm || !out) return fail(P3D_ERR_INVALID, "NULL argument");
    if (dtype != P3D_C64 && dtype != P3D_F32) return fail(P3D_ERR_INVALID, "unknown dtype %d", dtype);
    if (prm->niter < 1) return fail(P3D_ERR_INVALID, "niter must be >= 1");
    const bool percentile = (prm->thresh_op & P3D_OP_PERCENTILE) != 0;
    const int base_op = prm->thresh_op & ~P3D_OP_PERCENTILE;
    if (base_op < P3D_OP_HARD || base_op > P3D_OP_GARROTE)
        return fail(P3D_ERR_UNSUPPORTED, "thresh_op %d is not implemented by the HIP kernels", prm->thresh_op);
    if (prm->version < P3D_VER_REGULAR || prm->version > P3D_VER_ADAPTIVE)
        return fail(P3D_ERR_INVALID, "unknown version %d", prm->version);
    HIP_TRY(hipSetDevice(p->device));

    const int niter = prm->niter;
    const RunSwitches sw = read_switches();
    const bool profile = (prm->flags & P3D_FLAG_PROFILE) != 0;
    const bool early = prm->eps > 0.0;
    const bool adaptive = prm->version == P3D_VER_ADAPTIVE;

    // device-side schedule, state and cost accumulators
    const size_t ntau = (size_t)nslices * niter;
    if (p->tau_cap < ntau) {
        if (p->tau) hipFree(p->tau);
        p->tau = nullptr;
        p->tau_cap = 0;
        HIP_TRY(hipMalloc((void**)&p->tau, sizeof(c32) * ntau));
        p->tau_cap = ntau;
    }
    const size_t nsum = (size_t)(niter + 1) * nslices;
    if (p->sums_cap < nsum) {
        if (p->sums) hipFree(p->sums);
        p->sums = nullptr;
        p->sums_cap = 0;
        HIP_TRY(hipMalloc((void**)&p->sums, sizeof(double) * nsum));
        p->sums_cap = nsum;
    }
    std::vector<c32> tau_f(ntau);
    for (size_t i = 0; i < ntau; ++i) tau_f[i] = tau_for_device(tau[2 * i], tau[2 * i + 1], prm->thresh_op == P3D_OP_HARD);
    std::vector<int> done_h(nslices, 0);
    if (active)
        for (int s = 0; s < nslices; ++s) done_h[s] = active[s] ? 0 : -1;
    HIP_TRY(hipMemcpyAsync(p->tau, tau_f.data(), sizeof(c32) * ntau, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->done, done_h.data(), sizeof(int) * nslices, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipMemsetAsync(p->sums, 0, sizeof(double) * nsum, p->stream));

    size_t nev = 0;
    if (profile) {
        const size_t need = 2 * (size_t)niter + 2;
        while (p->prof_events.size() < need) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            p->prof_events.push_back(e);
        }
    }
    auto stamp = [&]() -> hipError_t {
        if (!profile) return hipSuccess;
        return hipEventRecord(p->prof_events[nev++], p->stream);
    };

    HIP_TRY(hipEventRecord(p->ev0, p->stream));

    // The -percentile operators rank the moduli of the whole spectrum (np.percentile, POCS.py:43-57).  The fused passes can do it
    // where the column-blocked work buffer holds exactly the slice (no padding columns): the column pass is split into forward
    // transform | rank + threshold | inverse transform.  Otherwise:
    const bool pct_fused = percentile && !p->generic && !sw.no_pct_fused && wk_slice_stride(p->nil, p->nxl) == p->slice_elems();
    if (percentile && !p->generic && !pct_fused) {
        // the tuned kernels never materialise the spectrum; ranking it needs the unfused pipeline -> a second, generic plan
        if (!p->pct_plan) {
            p3d_plan* q = nullptr;
            const int prc = create_plan(&q, p->device, p->nil, p->nxl, p->max_slices, true);
            if (prc) return prc;
            p->pct_plan = q;
        }
        return p3d_pocs_run_dev(p->pct_plan, x, dtype, mask, tau, active, prm, out, nslices, niter_done, sums, elapsed_ms);
    }
    if (p->generic) {
        // unfused any-length pipeline: first input, then per iteration fft2 -> threshold -> ifft2 -> re-insertion
        const size_t per_slice = p->slice_elems();
        bool any_off = early;
        for (int s = 0; s < nslices; ++s) any_off = any_off || done_h[s] != 0;
        const int* done_d = any_off ? p->done : nullptr;
        HIP_TRY(hipEventRecord(p->ev0, p->stream));
        HIP_TRY(gen_launch_update(p->work, x, dtype, mask, out, p->sums, 0, adaptive ? 1 : 0, 0, (float)prm->alpha, nslices, per_slice,
                                  done_d, 0, p->stream));
        for (int k = 0; k < niter; ++k) {
            const bool last = k + 1 == niter;
            if ((rc = gen_fft2(p, p->work, p->work, nslices, 0, done_d))) return rc;
            if (percentile) {  // tau[s][k] <- np.percentile(|X_s|, perc_k): two order statistics + linear interpolation
                std::vector<unsigned> sel_h((size_t)nslices * 16, 0u);
                std::vector<float> frac_h(nslices);
                for (int s = 0; s < nslices; ++s) {
                    const double perc = tau[2 * ((size_t)s * niter + k)];
                    double pos = perc / 100.0 * (double)(per_slice - 1);
                    if (!(pos >= 0.0)) pos = 0.0;
                    if (pos > (double)(per_slice - 1)) pos = (double)(per_slice - 1);
                    const double fl = std::floor(pos);
                    sel_h[(size_t)s * 8] = (unsigned)fl;
                    sel_h[((size_t)nslices + s) * 8] = (unsigned)std::min(fl + 1.0, (double)(per_slice - 1));
                    frac_h[s] = (float)(pos - fl);
                }
                HIP_TRY(hipMemcpyAsync(p->pct_sel, sel_h.data(), sizeof(unsigned) * sel_h.size(), hipMemcpyHostToDevice, p->stream));
                HIP_TRY(hipMemcpyAsync(p->pct_frac, frac_h.data(), sizeof(float) * nslices, hipMemcpyHostToDevice, p->stream));
                HIP_TRY(hipMemsetAsync(p->pct_hist, 0, sizeof(unsigned) * 2048 * (size_t)nslices, p->stream));
                for (int which = 0; which < 2; ++which) {
                    unsigned* sel = p->pct_sel + (size_t)which * nslices * 8;
                    for (int level = 0; level < 3; ++level) {
                        HIP_TRY(gen_launch_pct_hist(p->work, per_slice, sel, p->pct_hist, level, nslices, p->stream));
                        HIP_TRY(gen_launch_pct_scan(sel, p->pct_hist, level, nslices, p->stream));
                    }
                }
                HIP_TRY(gen_launch_pct_tau(p->pct_sel, p->pct_sel + (size_t)nslices * 8, p->pct_frac, p->tau, niter, k, nslices, p->stream));
                HIP_TRY(hipStreamSynchronize(p->stream));  // sel_h / frac_h are reused next iteration
            }
            HIP_TRY(gen_launch_shrink(p->work, p->tau, niter, k, base_op, nslices, per_slice, done_d, p->stream));
            if ((rc = gen_fft2(p, p->work, p->work, nslices, 1, done_d))) return rc;
            HIP_TRY(gen_launch_update(p->work, x, dtype, mask, out, p->sums + (size_t)(k + 1) * nslices, 1, (adaptive && !last) ? 1 : 0,
                                      (early || last) ? 1 : 0, (float)prm->alpha, nslices, per_slice, p->done, last ? 1 : 0, p->stream));
            if (early) conv_kernel<<<(nslices + 255) / 256, 256, 0, p->stream>>>(p->sums, p->done, nslices, k, prm->eps);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(p->ev1, p->stream));
        HIP_TRY(hipMemcpyAsync(done_h.data(), p->done, sizeof(int) * nslices, hipMemcpyDeviceToHost, p->stream));
        if (sums) HIP_TRY(hipMemcpyAsync(sums, p->sums, sizeof(double) * nsum, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        if (niter_done)
            for (int s = 0; s < nslices; ++s) niter_done[s] = done_h[s] < 0 ? 0 : (done_h[s] > 0 ? done_h[s] : niter);
        if (elapsed_ms) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
            *elapsed_ms = ms;
        }
        p->prof_col_n = p->prof_row_n = 0;
        return P3D_OK;
    }

    // packed trace mask: binary masks (the workflow's fold-derived mask, cube_POCS_interpolation_3D.py:242-244)
    // travel as one 16-bit word per thread and row; anything else keeps the float weights
    const bool flex_rows = is_flex(p->ops_row);   // the flexible row pass reads the float weights
    int nonbinary = 0;
    unsigned nobs = 0;
    if (primed_in) {   // packed by p3d_pocs_prime_dev, still in place
        nonbinary = primed_state.nonbinary;
        nobs = primed_state.nobs;
        HIP_TRY(hipMemsetAsync(p->flag, 0, 2 * sizeof(int), p->stream));
    } else if ((rc = pack_mask(p, mask, &nonbinary, &nobs))) {
        return rc;
    }
    if (sw.no_mask_bits) nonbinary = 1;  // experiments only

    // Small slices (32 ... 128 points per axis, at most 8192 per slice): the whole job in ONE kernel, a workgroup per slice, the
    // slice in registers and LDS for all iterations (p3d_resident.hip).  Bit-identical to the passes below, which stay for the
    // options it does not cover (APOCS, non-binary masks) and behind P3D_NO_RESIDENT=1.
    if (!nonbinary && !adaptive && !percentile && !sw.no_resident && !flex_rows && !is_flex(p->ops_col) && resident_supported(p->nil, p->nxl) &&
        (size_t)p->nil * p->nxl <= RESIDENT_MAX_POINTS) {
        ResidentArgs ra{};
        ra.x = x; ra.out = out; ra.bits = p->bits; ra.tau = p->tau; ra.done = p->done; ra.sums = p->sums;
        ra.tw_row = p->tw_row; ra.tw_col = p->tw_col;
        ra.nslices = nslices; ra.niter = niter; ra.op = base_op; ra.dtype = dtype;
        ra.alpha = (float)prm->alpha; ra.scale = (float)(1.0 / ((double)p->nil * (double)p->nxl)); ra.eps = prm->eps;
        HIP_TRY(resident_launch(p->nil, p->nxl, ra, p->stream));
        HIP_TRY(hipEventRecord(p->ev1, p->stream));
        HIP_TRY(hipMemcpyAsync(done_h.data(), p->done, sizeof(int) * nslices, hipMemcpyDeviceToHost, p->stream));
        if (sums) HIP_TRY(hipMemcpyAsync(sums, p->sums, sizeof(double) * nsum, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        if (niter_done)
            for (int s = 0; s < nslices; ++s) niter_done[s] = done_h[s] < 0 ? 0 : (done_h[s] > 0 ? done_h[s] : niter);
        if (elapsed_ms) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
            *elapsed_ms = ms;
        }
        p->last_nonzero_fraction = -1.0;
        p->prof_col_n = p->prof_row_n = 0;
        return P3D_OK;
    }
    // Compact observed samples for the steady-state row pass: only the observed positions of x are non-zero in
    // the workflow (x = stacked traces, mask = fold >= 1); ROW_FIRST verifies that and the full cube is used if not.
    bool compact = want_compact(p, nonbinary, nobs, niter, sw);
    if (compact && (rc = ensure_xc(p, nslices, nobs, dtype))) return rc;

    RowArgs r = row_args(p, nslices);
    r.x = x;
    r.mask = nonbinary ? mask : nullptr;
    r.bits = nonbinary ? nullptr : p->bits;
    // APOCS runs the wave-uniform persistent pass too (round 3): the first pass, row_kernel with the input mix, writes the compact array
    // in the word order when it is handed the tables; with the early exit it stores every iterate there (write_out below).
    r.bits64 = nonbinary ? nullptr : p->bits64;
    r.cbase = nonbinary ? nullptr : p->cbase;
    r.bits32 = (nonbinary || !p->use32) ? nullptr : p->bits32;
    r.xc = compact ? p->xc : nullptr;
    r.rowbase = p->rowbase;
    r.nobs = nobs;
    r.violation = p->flag + 1;
    r.work = p->work;
    r.out = out;
    // the per-slice state array is only consulted when a slice can actually be switched off
    bool any_off = early;
    for (int s = 0; s < nslices; ++s) any_off = any_off || done_h[s] != 0;
    r.sums = p->rowsum;
    r.done = any_off ? p->done : nullptr;
    r.dtype = dtype;
    r.real_2048 = sw.real_2048 ? 1 : 0;
    r.tstore = sw.no_tstore ? 0 : 1;
    r.adaptive = adaptive ? 1 : 0;
    // Early exit: a slice that converges at iteration k leaves the forward row transform of its iterate in the work buffer;
    // a "finalize" launch after the convergence test turns that back into `out` for exactly those slices, so the steady
    // state does not store every iterate of every slice (8 B/point per iteration).  APOCS feeds a mix of iterate and
    // observation forward instead of the iterate: there the per-iteration store stays.
    const bool finalize = early && !adaptive;
    r.write_out = (early && !finalize) ? 1 : 0;
    r.alpha = (float)prm->alpha;
    r.sum_row = 0;
    // rows of finished / empty slices are skipped by the kernels: their partial sums must read as zero
    HIP_TRY(hipMemsetAsync(p->rowsum, 0, sizeof(double) * (size_t)p->nil * nslices, p->stream));
    // Real cubes with the hard operator: the spectrum stays Hermitian, so row pairs share one complex transform and the work
    // buffer holds half the columns (row_real_kernel).  Needs the compact observed samples and the lane-mask tables.
    // (The flexible-length row pass keeps the pair in LDS and takes any real mask: no compact samples, no tables needed there.)
    bool real_path = dtype == P3D_F32 && base_op == P3D_OP_HARD && !adaptive && !percentile && p->ops_row->row_real != nullptr && !sw.no_real &&
                     (flex_rows ? p->nil % 2 == 0 : (compact && r.bits64 && r.cbase && p->pipe_wgs > 0));
    // (a primed real first pass found no energy at unobserved positions -- p3d_pocs_prime_dev falls back to the complex pass otherwise)
    const bool primed_real = primed_in && primed_state.real && real_path;
    if (real_path && !primed_real) {
        const hipError_t re = p->ops_row->row_real(REAL_FIRST, r, p->pipe_wgs, p->stream);
        if (re == hipErrorNotSupported) real_path = false;
        else HIP_TRY(re);
    }
    // Rows on the mixed-radix register engine (p3d_mix.hpp) with a binary mask: the packed words instead of the float weights, and in the
    // steady state the compact observed samples its first pass writes (thread-major order, RowArgs::mbase) -- 1 bit + 8 (1 - missing) bytes
    // per point and iteration instead of 12.
    bool mix_compact = false;
    if (flex_rows && p->mix_binary && !real_path && !sw.no_mask_bits) {
        r.mbits = p->mbits;
        r.mbase = p->mbase;
        mix_compact = niter > 1 && !sw.no_compact && nobs > 0 && (double)nobs < 0.75 * (double)p->slice_elems();
        if (mix_compact) {
            if ((rc = ensure_xc(p, nslices, nobs, dtype))) return rc;
            r.xc = p->xc;
        }
    }
    // The primed first pass is the one this run would make itself (row pairs for the real path, the complex one without the APOCS input
    // mix otherwise): work buffer, compact samples and sum |x_obs| are there.
    const bool primed = primed_in && !adaptive && (real_path ? primed_real : !primed_state.real);
    if (primed) {
        HIP_TRY(hipMemcpyAsync(p->sums, p->sum0, sizeof(double) * nslices, hipMemcpyDeviceToDevice, p->stream));
        // the primed pass knew nothing of `active`: a slice the caller switched off reports sums[0] = 0, as on the unprimed path
        if (any_off) zero_off_sums_kernel<<<(nslices + 255) / 256, 256, 0, p->stream>>>(p->sums, p->done, nslices);
        if (primed_state.violation) r.xc = nullptr;
    } else if (!real_path) {
        HIP_TRY(first_row_pass(p, r));
    }
    if ((compact || mix_compact) && !primed) {  // did every unobserved position hold a zero?
        int violation = 0;
        HIP_TRY(hipMemcpyAsync(&violation, p->flag + 1, sizeof(int), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        if (violation) {
            r.xc = nullptr;
            if (real_path) {   // the row-pair passes live on the compact samples: take the complex path from the start
                real_path = false;
                HIP_TRY(hipMemsetAsync(p->rowsum, 0, sizeof(double) * (size_t)p->nil * nslices, p->stream));
                HIP_TRY(first_row_pass(p, r));
            }
        }
    }
    if (!primed) reduce_rows_kernel<<<nslices, 256, 0, p->stream>>>(p->rowsum, p->sums, p->nil);

    ColArgs c = col_args(p, nslices);
    c.in = p->work;
    c.out = p->work;
    c.tau = p->tau;
    c.done = any_off ? p->done : nullptr;
    c.niter = niter;
    c.op = base_op;
    // tiles of the spectrum that the threshold empties are neither transformed back, stored nor read again
    const bool sparse = p->sparse_ok && !sw.no_sparse && !percentile;   // (a percentile keeps a fixed share of the coefficients)
    const int nblocks = (p->nxl + 7) / 8, groups = flex_rows ? 1 : p->ops_row->tpl / 8;
    const int col_t = is_flex(p->ops_col) ? flex_col_tile(p->nil) : p->ops_col->col_tile;
    if (sparse) {
        c.nzflag = p->nzflag;
        r.zero_off = (unsigned)(wk_slice_stride(p->nil, p->nxl) * (size_t)p->max_slices);
        HIP_TRY(hipMemsetAsync(p->nzcount, 0, sizeof(unsigned long long), p->stream));
    }
    // The persistent column pass (col_pipe_kernel) wins where nearly all tiles end at the threshold (-4 ... -12 % by shape at 4 % kept
    // blocks) and loses as soon as a noticeable share is transformed back and stored (+14 % at the 6 % of a 20-iteration job, +9 % dense:
    // profiles/r02_colpass_persistent.txt).  Which of the two a job is cannot be known before it has run, and choosing from the plan's
    // previous job cost the 20-iteration bench line 7 %: there it stays an experiment behind P3D_FORCE_COLPIPE=1, bit-identical and tested.
    // Columns of 2048 points are another matter: their 8-column tile takes a whole CU (1024 threads, 157 KiB of LDS), so the one-launch
    // pass has nothing to overlap a tile's loads with, and the persistent pass wins at every density measured (-5 % at 8 iterations,
    // -10 ... -12 % at 20 ... 60): the default there (P3D_NO_COLPIPE=1 switches it off).
    const bool colpipe = sparse && p->cus > 0 && p->ops_col->col_pipe != nullptr &&
                         (sw.force_colpipe || (p->ops_col->n >= 2048 && !sw.no_colpipe));
    p->last_nonzero_fraction = -1.0;
    const int n2_work = real_path ? p->nxl / 2 + 1 : p->nxl;   // columns of the work buffer
    const int tiles_work = (n2_work + col_t - 1) / col_t;
    c.n2 = n2_work;

    // -percentile operators: ranks and interpolation weights of np.percentile for every iteration (tau holds the percentages)
    std::vector<unsigned> pct_sel_h;
    std::vector<float> pct_frac_h;
    if (percentile) {
        const size_t per_slice = p->slice_elems();
        if (!p->pct_sel) {
            HIP_TRY(hipMalloc((void**)&p->pct_sel, sizeof(unsigned) * 16 * (size_t)p->max_slices));
            HIP_TRY(hipMalloc((void**)&p->pct_hist, sizeof(unsigned) * 2048 * (size_t)p->max_slices));
            HIP_TRY(hipMalloc((void**)&p->pct_frac, sizeof(float) * (size_t)p->max_slices));
        }
        pct_sel_h.assign((size_t)niter * nslices * 16, 0u);
        pct_frac_h.assign((size_t)niter * nslices, 0.f);
        for (int k = 0; k < niter; ++k)
            for (int s = 0; s < nslices; ++s) {
                const double perc = tau[2 * ((size_t)s * niter + k)];
                double pos = perc / 100.0 * (double)(per_slice - 1);
                if (!(pos >= 0.0)) pos = 0.0;
                if (pos > (double)(per_slice - 1)) pos = (double)(per_slice - 1);
                const double fl = std::floor(pos);
                unsigned* sel = &pct_sel_h[(size_t)k * nslices * 16];
                sel[(size_t)s * 8] = (unsigned)fl;
                sel[((size_t)nslices + s) * 8] = (unsigned)std::min(fl + 1.0, (double)(per_slice - 1));
                pct_frac_h[(size_t)k * nslices + s] = (float)(pos - fl);
            }
        HIP_TRY(hipMemsetAsync(p->pct_hist, 0, sizeof(unsigned) * 2048 * (size_t)nslices, p->stream));
    }

    HIP_TRY(stamp());
    int last_finalized = 0;   // early exit: slices with done <= last_finalized have been handed to `out`
    // Debug mode of the invariant the deferred finalize rests on (RowArgs::only_done_lo, ColArgs::done): the work slice of a converged slice
    // that has not been handed back yet must not change.  One blocking round trip per iteration: tests only (tests/test_gpu_parity.py).
    const bool check_done_rows = finalize && getenv("P3D_CHECK_DONE_ROWS") != nullptr;
    std::vector<unsigned long long> done_sum_first(check_done_rows ? nslices : 0, 0ull), done_sum_now(check_done_rows ? nslices : 0, 0ull);
    std::vector<char> done_sum_seen(check_done_rows ? nslices : 0, 0);
    unsigned long long* done_sum_d = nullptr;
    if (check_done_rows) HIP_TRY(hipMalloc((void**)&done_sum_d, sizeof(unsigned long long) * nslices));
    struct FreeSum { unsigned long long* p; ~FreeSum() { if (p) hipFree(p); } } free_sum{done_sum_d};
    for (int k = 0; k < niter; ++k) {
        c.iter = k;
        if (percentile) {
            // forward column transform | rank |X|, tau_k = the interpolated order statistic, threshold | inverse column transform
            const size_t per_slice = p->slice_elems();
            ColArgs cf = c;
            cf.tau = nullptr;
            cf.nzflag = nullptr;
            HIP_TRY(p->ops_col->col(COL_FWD, cf, p->stream));
            HIP_TRY(hipMemcpyAsync(p->pct_sel, &pct_sel_h[(size_t)k * nslices * 16], sizeof(unsigned) * 16 * (size_t)nslices, hipMemcpyHostToDevice, p->stream));
            HIP_TRY(hipMemcpyAsync(p->pct_frac, &pct_frac_h[(size_t)k * nslices], sizeof(float) * nslices, hipMemcpyHostToDevice, p->stream));
            for (int which = 0; which < 2; ++which) {
                unsigned* sel = p->pct_sel + (size_t)which * nslices * 8;
                for (int level = 0; level < 3; ++level) {
                    HIP_TRY(gen_launch_pct_hist(p->work, per_slice, sel, p->pct_hist, level, nslices, p->stream));
                    HIP_TRY(gen_launch_pct_scan(sel, p->pct_hist, level, nslices, p->stream));
                }
            }
            HIP_TRY(gen_launch_pct_tau(p->pct_sel, p->pct_sel + (size_t)nslices * 8, p->pct_frac, p->tau, niter, k, nslices, p->stream));
            HIP_TRY(gen_launch_shrink(p->work, p->tau, niter, k, base_op, nslices, per_slice, c.done, p->stream));
            HIP_TRY(p->ops_col->col(COL_INV, cf, p->stream));
        } else {
            hipError_t ce = hipErrorNotSupported;
            if (colpipe) ce = p->ops_col->col_pipe(c, p->cus, p->stream);
            if (ce == hipErrorNotSupported) ce = p->ops_col->col(COL_ITER, c, p->stream);
            HIP_TRY(ce);
        }
        if (sparse && real_path && !flex_rows) {
            nz_real_kernel<<<(nslices * 16 * (p->ops_row->tpl >= 64 ? p->ops_row->tpl / 64 : 1) + 3) / 4, 256, 0, p->stream>>>(p->nzflag, p->nzl, p->nzcount, nslices, tiles_work, col_t, p->nxl, p->ops_row->tpl, c.done);
            r.nzl = p->nzl;
        } else if (sparse && flex_rows) {   // the flexible row pass reads the tile flags themselves; count the kept blocks for the statistics
            const int nb_work = (n2_work + 7) / 8;   // the half spectrum of a real cube has its own tile count
            nz_count_kernel<<<(nslices * nb_work + 255) / 256, 256, 0, p->stream>>>(p->nzflag, p->nzcount, nslices, tiles_work, col_t, nb_work, c.done);
            r.nzflag = p->nzflag;
            r.nz_tiles = tiles_work;
            r.nz_col_t = col_t;
        } else if (sparse) {
            nz_pack_kernel<<<(nslices * groups + 255) / 256, 256, 0, p->stream>>>(p->nzflag, p->nzm, p->nzcount, nslices, p->tiles, col_t, groups, nblocks,
                                                                                c.done, p->nzl);
            r.nzm = p->nzm;
            r.nzl = p->nzl;
        }
        HIP_TRY(stamp());
        r.sum_row = k + 1;
        bool piped = false;
        // rows of 1024 samples whose first pass wrote the compact samples in row_pipe32_kernel's order: that family all the way
        // (r.xc is null whenever the first pass wrote none: non-binary mask, no compact samples wanted, energy at unobserved traces)
        const bool fam32 = !real_path && p->use32 && r.xc != nullptr && r.bits32 != nullptr;
        if (real_path) {
            HIP_TRY(p->ops_row->row_real(k + 1 < niter ? REAL_MID : REAL_LAST, r, p->pipe_wgs, p->stream));
            piped = true;
        } else if (fam32) {
            const hipError_t fe = p->ops_row->row_pipe32(k + 1 < niter ? PIPE_MID : PIPE_LAST, r, p->pipe_wgs, p->stream);
            if (fe == hipErrorNotSupported) return fail(P3D_ERR_HIP, "row_pipe32_kernel refused a job its first pass had accepted");
            HIP_TRY(fe);
            piped = true;
        } else if (k + 1 < niter && p->pipe_wgs > 0) {  // steady state: persistent, software-pipelined row pass
            const hipError_t pe = p->ops_row->row_pipe(r, p->pipe_wgs, p->stream);
            if (pe == hipSuccess) piped = true;
            else if (pe != hipErrorNotSupported) HIP_TRY(pe);
        }
        if (!piped && k + 1 == niter && p->pipe_wgs > 0 && p->bits64 != nullptr && p->ops_row->row_pipe64 != nullptr) {
            const hipError_t le = p->ops_row->row_pipe64(PIPE_LAST, r, p->pipe_wgs, p->stream);   // last pass: compact samples in, whole rows out
            if (le == hipSuccess) piped = true;
            else if (le != hipErrorNotSupported) HIP_TRY(le);
        }
        if (!piped) HIP_TRY(p->ops_row->row(k + 1 < niter ? ROW_MID : ROW_LAST, r, p->stream));
        HIP_TRY(stamp());
        reduce_rows_kernel<<<nslices, 256, 0, p->stream>>>(p->rowsum, p->sums + (size_t)(k + 1) * nslices, p->nil);
        if (early) {
            conv_kernel<<<(nslices + 255) / 256, 256, 0, p->stream>>>(p->sums, p->done, nslices, k, prm->eps);
            // slices that just finished are skipped from now on: their rows must read as zero afterwards
            HIP_TRY(hipMemsetAsync(p->rowsum, 0, sizeof(double) * (size_t)p->nil * nslices, p->stream));
            // (the finalize launch -- mostly workgroups that find nothing to do -- runs every FIN_EVERY iterations and before the last pass: a slice that has
            // converged keeps its work rows, nothing touches them any more)
            constexpr int FIN_EVERY = 8;
            if (check_done_rows) {
                done_rows_checksum_kernel<<<nslices, 256, 0, p->stream>>>(reinterpret_cast<const unsigned long long*>(p->work), wk_slice_stride(p->nil, n2_work),
                                                                         p->done, last_finalized, done_sum_d);
                HIP_TRY(hipMemcpyAsync(done_sum_now.data(), done_sum_d, sizeof(unsigned long long) * nslices, hipMemcpyDeviceToHost, p->stream));
                HIP_TRY(hipMemcpyAsync(done_h.data(), p->done, sizeof(int) * nslices, hipMemcpyDeviceToHost, p->stream));
                HIP_TRY(hipStreamSynchronize(p->stream));
                for (int s = 0; s < nslices; ++s) {
                    if (done_h[s] <= last_finalized) continue;   // running, empty or already handed back
                    if (!done_sum_seen[s]) { done_sum_seen[s] = 1; done_sum_first[s] = done_sum_now[s]; }
                    else if (done_sum_first[s] != done_sum_now[s])
                        return fail(P3D_ERR_HIP, "P3D_CHECK_DONE_ROWS: the work rows of slice %d (converged at iteration %d) changed before its finalize launch (iteration %d)", s,
                                    done_h[s], k + 1);
                }
            }
            if (finalize && k + 1 < niter && (k + 1 - last_finalized >= FIN_EVERY || k + 2 >= niter)) {
                RowArgs f = r;
                f.nzm = nullptr;  // the rows it reads were written by the row pass: all there
                f.nzl = nullptr;
                f.nzflag = nullptr;
                f.only_done_lo = last_finalized;
                f.only_done = k + 1;
                last_finalized = k + 1;
                f.plain = 0;      // the observed samples are needed (exact hand-back at observed traces)
                f.sums = nullptr;
                f.done = p->done;
                f.scale = (float)(1.0 / (double)p->nxl);   // one row transform to undo, not a 2-D one
                if (real_path) HIP_TRY(p->ops_row->row_real(REAL_LAST, f, p->pipe_wgs, p->stream));
                else HIP_TRY(p->ops_row->row(ROW_LAST, f, p->stream));
            }
        }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(p->ev1, p->stream));

    HIP_TRY(hipMemcpyAsync(done_h.data(), p->done, sizeof(int) * nslices, hipMemcpyDeviceToHost, p->stream));
    if (sums) HIP_TRY(hipMemcpyAsync(sums, p->sums, sizeof(double) * nsum, hipMemcpyDeviceToHost, p->stream));
    unsigned long long kept_blocks = 0;
    if (sparse) HIP_TRY(hipMemcpyAsync(&kept_blocks, p->nzcount, sizeof kept_blocks, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (sparse) p->last_nonzero_fraction = (double)kept_blocks / ((double)niter * nslices * (real_path ? (n2_work + 7) / 8 : nblocks));

    if (niter_done)
        for (int s = 0; s < nslices; ++s) niter_done[s] = done_h[s] < 0 ? 0 : (done_h[s] > 0 ? done_h[s] : niter);
    if (elapsed_ms) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
        *elapsed_ms = ms;
    }
    if (profile) {
        p->prof_col_ms = p->prof_row_ms = 0;
        p->prof_col_n = p->prof_row_n = 0;
        for (int k = 0; k < niter; ++k) {
            float a = 0.f, b = 0.f;
            HIP_TRY(hipEventElapsedTime(&a, p->prof_events[2 * k], p->prof_events[2 * k + 1]));
            HIP_TRY(hipEventElapsedTime(&b, p->prof_events[2 * k + 1], p->prof_events[2 * k + 2]));
            p->prof_col_ms += a;
            p->prof_col_n += 1;
            if (k + 1 < niter) {  // the last space pass is ROW_LAST (no forward transform): not averaged in
                p->prof_row_ms += b;
                p->prof_row_n += 1;
            }
        }
    }
    return P3D_OK;
}

int p3d_pocs_run(p3d_plan* p, const void* x, int dtype, const float* mask, const double* tau, const uint8_t* active,
                 const p3d_pocs_params* prm, void* out, int nslices, int32_t* niter_done, double* sums,
                 double* elapsed_ms)
{
    int rc = check_batch(p, nslices);
    if (rc) return rc;
    if (!x || !mask || !out) return fail(P3D_ERR_INVALID, "NULL argument");
    if (dtype != P3D_C64 && dtype != P3D_F32) return fail(P3D_ERR_INVALID, "unknown dtype %d", dtype);
    HIP_TRY(hipSetDevice(p->device));
    if ((rc = ensure_staging(p, sizeof(c32) * p->slice_elems() * p->max_slices))) return rc;
    const size_t esz = dtype == P3D_C64 ? sizeof(c32) : sizeof(float);
    const size_t bytes = esz * p->slice_elems() * nslices;
    HIP_TRY(hipMemcpy(p->st_x, x, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->st_mask, mask, sizeof(float) * p->slice_elems(), hipMemcpyHostToDevice));
    if ((rc = p3d_pocs_run_dev(p, p->st_x, dtype, p->st_mask, tau, active, prm, p->st_out, nslices, niter_done, sums,
                               elapsed_ms)))
        return rc;
    HIP_TRY(hipMemcpy(out, p->st_out, bytes, hipMemcpyDeviceToHost));
    return P3D_OK;
}

namespace {
struct DevBuf {  // frees on scope exit
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
};
}  // namespace

static int helper_check(int device, size_t ntraces, int nfft)
{
    if (ntraces < 1 || nfft < 1) return fail(P3D_ERR_INVALID, "ntraces and nfft must be positive");
    if (nfft > GEN_MAX_N || gen_make_plan(nfft).nf < 0) return fail(P3D_ERR_UNSUPPORTED, "nfft = %d: lengths up to %d are supported", nfft, GEN_MAX_N);
    if (ntraces > 2147483647u) return fail(P3D_ERR_INVALID, "too many traces for one call: split the cube");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(P3D_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    return P3D_OK;
}

static double fft_freq(int k, int nfft, double dt) { return (k < (nfft + 1) / 2 ? k : k - nfft) / (nfft * dt); }  // np.fft.fftfreq

// One transform along axis 0 of a row-major [nfft][ntr] complex matrix on the device, in place.  Lengths with tuned or flexible
// column kernels run those (eight traces per workgroup, 64-byte pieces); longer ones the line-per-workgroup fallback.
static int axis0_fft(int device, c32* work, int nfft, size_t ntr, int inverse)
{
    // (the column kernels address a row-major matrix with 32-bit element offsets)
    const LineOps* ops = (double)nfft * (double)ntr < 4294967296.0 ? find_ops(nfft) : nullptr;
    if (!ops) {
        const GenPlan pl = gen_make_plan(nfft);
        std::vector<c32> tw(nfft);
        gen_build_twiddles(nfft, tw.data());
        DevBuf dtw;
        HIP_TRY(hipMalloc(&dtw.p, sizeof(c32) * nfft));
        HIP_TRY(hipMemcpy(dtw.p, tw.data(), sizeof(c32) * nfft, hipMemcpyHostToDevice));
        HIP_TRY(gen_launch_line_fft(work, work, (const c32*)dtw.p, pl, inverse ? INV : FWD, 1.0f, 1, nfft, (int)ntr, false, nullptr, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        return P3D_OK;
    }
    std::vector<c32> host;
    if (is_flex(ops)) {
        flex_build_table(nfft, host);
    } else {
        host.resize(ops->col_tw_slots);
        ops->build_col_tw(host.data());
    }
    DevBuf dtw;
    HIP_TRY(hipMalloc(&dtw.p, sizeof(c32) * host.size()));
    HIP_TRY(hipMemcpy(dtw.p, host.data(), sizeof(c32) * host.size(), hipMemcpyHostToDevice));
    ColArgs c{};
    c.tw = (const c32*)dtw.p;
    c.in = work;
    c.out = work;
    c.n2 = (int)ntr;
    c.nslices = 1;
    c.len = nfft;
    c.in_std = c.out_std = 1;
    HIP_TRY(hipDeviceGetAttribute(&c.cus, hipDeviceAttributeMultiprocessorCount, device));   // (no plan here: the device the caller named)
    HIP_TRY(ops->col(inverse ? COL_INV : COL_FWD, c, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    return P3D_OK;
}

int p3d_time2freq_dev(int device, const float* x, int nt, size_t ntr, double dt, double t0, int nfft, int real_only, const float* window,
                      void* out)
{
    if (!x || !out) return fail(P3D_ERR_INVALID, "NULL buffer");
    if (nt < 1 || nfft < nt) return fail(P3D_ERR_INVALID, "need 1 <= nt <= nfft");
    int rc = helper_check(device, ntr, nfft);
    if (rc) return rc;
    const int nfreq = real_only ? nfft / 2 + 1 : nfft;
    std::vector<c32> fac(nfreq);
    for (int k = 0; k < nfreq; ++k) {
        // rfft bins are the non-negative frequencies k/(nfft*dt) (np.fft.rfftfreq), also for k = nfft/2
        const double f = real_only ? k / (nfft * dt) : fft_freq(k, nfft, dt);
        const double ang = -6.283185307179586476925286766559 * f * t0, w = window ? (double)window[k] : 1.0;
        fac[k] = c32{(float)(dt * w * std::cos(ang)), (float)(dt * w * std::sin(ang))};
    }
    DevBuf dwork, dfac;
    HIP_TRY(hipMalloc(&dwork.p, sizeof(c32) * (size_t)nfft * ntr));
    HIP_TRY(hipMalloc(&dfac.p, sizeof(c32) * nfreq));
    HIP_TRY(hipMemcpy(dfac.p, fac.data(), sizeof(c32) * nfreq, hipMemcpyHostToDevice));
    c32* work = (c32*)dwork.p;
    HIP_TRY(gen_launch_t2f_pad(x, work, nt, nfft, ntr, nullptr));
    if ((rc = axis0_fft(device, work, nfft, ntr, 0))) return rc;
    HIP_TRY(gen_launch_scale_rows(work, (c32*)out, (const c32*)dfac.p, nfreq, ntr, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    return P3D_OK;
}

int p3d_time2freq(int device, const float* x, int nt, size_t ntr, double dt, double t0, int nfft, int real_only, const float* window,
                  void* out)
{
    if (!x || !out) return fail(P3D_ERR_INVALID, "NULL buffer");
    if (nt < 1 || nfft < nt) return fail(P3D_ERR_INVALID, "need 1 <= nt <= nfft");
    int rc = helper_check(device, ntr, nfft);
    if (rc) return rc;
    const int nfreq = real_only ? nfft / 2 + 1 : nfft;
    DevBuf dx, dout;
    HIP_TRY(hipMalloc(&dx.p, sizeof(float) * (size_t)nt * ntr));
    HIP_TRY(hipMalloc(&dout.p, sizeof(c32) * (size_t)nfreq * ntr));
    HIP_TRY(hipMemcpy(dx.p, x, sizeof(float) * (size_t)nt * ntr, hipMemcpyHostToDevice));
    if ((rc = p3d_time2freq_dev(device, (const float*)dx.p, nt, ntr, dt, t0, nfft, real_only, window, dout.p))) return rc;
    HIP_TRY(hipMemcpy(out, dout.p, sizeof(c32) * (size_t)nfreq * ntr, hipMemcpyDeviceToHost));
    return P3D_OK;
}

int p3d_freq2time_dev(int device, const void* X, int nfreq, const int32_t* kidx, size_t ntr, double dt, double t0, int nfft, int real_only,
                      float* out)
{
    if (!X || !out || !kidx) return fail(P3D_ERR_INVALID, "NULL buffer");
    if (nfreq < 1 || nfreq > nfft) return fail(P3D_ERR_INVALID, "need 1 <= nfreq <= nfft");
    int rc = helper_check(device, ntr, nfft);
    if (rc) return rc;
    std::vector<c32> fac(nfft, c32{0.f, 0.f});
    std::vector<int> src(nfft, INT_MIN);
    for (int r = 0; r < nfreq; ++r) {
        const int k = kidx[r];
        if (k < 0 || k >= nfft) return fail(P3D_ERR_INVALID, "kidx[%d] = %d outside 0..nfft-1", r, k);
        src[k] = r;
        if (real_only && k > 0 && 2 * k != nfft) src[nfft - k] = -(r + 1);  // Hermitian partner: conj(X[k])
    }
    for (int k = 0; k < nfft; ++k) {
        // undo the phase of the forward transform; for the mirrored bins conj(X) carries exp(+i..), undone by its own f_k
        // (the half spectrum stores the Nyquist bin at +f_nyq, np.fft.rfftfreq; the full one at -f_nyq, np.fft.fftfreq)
        const double f = (real_only && 2 * k == nfft) ? k / (nfft * dt) : fft_freq(k, nfft, dt);
        const double ang = 6.283185307179586476925286766559 * f * t0;
        fac[k] = c32{(float)std::cos(ang), (float)std::sin(ang)};
    }
    DevBuf dwork, dfac, dsrc;
    HIP_TRY(hipMalloc(&dwork.p, sizeof(c32) * (size_t)nfft * ntr));
    HIP_TRY(hipMalloc(&dfac.p, sizeof(c32) * nfft));
    HIP_TRY(hipMalloc(&dsrc.p, sizeof(int) * nfft));
    HIP_TRY(hipMemcpy(dfac.p, fac.data(), sizeof(c32) * nfft, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dsrc.p, src.data(), sizeof(int) * nfft, hipMemcpyHostToDevice));
    c32* work = (c32*)dwork.p;
    HIP_TRY(gen_launch_f2t_fill((const c32*)X, work, (const c32*)dfac.p, (const int*)dsrc.p, nfft, ntr, nullptr));
    if ((rc = axis0_fft(device, work, nfft, ntr, 1))) return rc;
    // true_amplitude: the inverse carries 1/(nfft*dt)
    HIP_TRY(gen_launch_real_part(work, out, (size_t)nfft * ntr, (float)(1.0 / (nfft * dt)), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    return P3D_OK;
}

int p3d_freq2time(int device, const void* X, int nfreq, const int32_t* kidx, size_t ntr, double dt, double t0, int nfft, int real_only,
                  float* out)
{
    if (!X || !out || !kidx) return fail(P3D_ERR_INVALID, "NULL buffer");
    if (nfreq < 1 || nfreq > nfft) return fail(P3D_ERR_INVALID, "need 1 <= nfreq <= nfft");
    int rc = helper_check(device, ntr, nfft);
    if (rc) return rc;
    DevBuf dX, dout;
    HIP_TRY(hipMalloc(&dX.p, sizeof(c32) * (size_t)nfreq * ntr));
    HIP_TRY(hipMalloc(&dout.p, sizeof(float) * (size_t)nfft * ntr));
    HIP_TRY(hipMemcpy(dX.p, X, sizeof(c32) * (size_t)nfreq * ntr, hipMemcpyHostToDevice));
    if ((rc = p3d_freq2time_dev(device, dX.p, nfreq, kidx, ntr, dt, t0, nfft, real_only, (float*)dout.p))) return rc;
    HIP_TRY(hipMemcpy(out, dout.p, sizeof(float) * (size_t)nfft * ntr, hipMemcpyDeviceToHost));
    return P3D_OK;
}

int p3d_last_sparsity(p3d_plan* p, double* nonzero_fraction)
{
    if (!p || !nonzero_fraction) return fail(P3D_ERR_INVALID, "NULL argument");
    *nonzero_fraction = p->last_nonzero_fraction;
    return P3D_OK;
}

int p3d_last_profile(p3d_plan* p, double* col_ms, int* col_n, double* row_ms, int* row_n)
{
    if (!p) return fail(P3D_ERR_INVALID, "NULL plan");
    if (col_ms) *col_ms = p->prof_col_n ? p->prof_col_ms / p->prof_col_n : 0.0;
    if (col_n) *col_n = p->prof_col_n;
    if (row_ms) *row_ms = p->prof_row_n ? p->prof_row_ms / p->prof_row_n : 0.0;
    if (row_n) *row_n = p->prof_row_n;
    return P3D_OK;
}

}  // extern "C"


// ---- several devices from one process ----------------------------------------------------------------------------------------
// Slices are independent (cube_POCS_interpolation_3D.py:314-340 hands them to a farm of workers): contiguous blocks of the slice
// axis go to the listed devices, one host thread and one plan per entry, each block in chunks of ~256 MiB.  The blocks are host
// arrays, so every device copies its own results back and no collective is needed.  (One process per GPU with torch.distributed
// -- sharding.py, bench.py -- is the other way to shard; this entry point serves callers that stay in one process.)
namespace {
struct BlockJob { int dev, lo, hi; int rc = P3D_OK; std::string err; };

template <class F>
int run_blocks(int ndev, const int* devices, int nslices, F&& body)
{
    if (ndev < 1 || !devices) return fail(P3D_ERR_INVALID, "no devices");
    if (nslices < 1) return fail(P3D_ERR_INVALID, "nslices = %d", nslices);
    std::vector<BlockJob> jobs((size_t)ndev);
    for (int d = 0; d < ndev; ++d) {   // the same split as sharding.slice_block
        const int base = nslices / ndev, rem = nslices % ndev;
        jobs[d].dev = devices[d];
        jobs[d].lo = d * base + (d < rem ? d : rem);
        jobs[d].hi = jobs[d].lo + base + (d < rem ? 1 : 0);
    }
    std::vector<std::thread> threads;
    for (auto& j : jobs)
        if (j.hi > j.lo) threads.emplace_back([&j, &body]() {
            j.rc = body(j);
            if (j.rc != P3D_OK) j.err = p3d_last_error();   // the message lives in this thread's storage
        });
    for (auto& t : threads) t.join();
    for (const auto& j : jobs)
        if (j.rc != P3D_OK) return fail(j.rc, "device %d, slices %d..%d: %s", j.dev, j.lo, j.hi - 1, j.err.c_str());
    return P3D_OK;
}
int chunk_slices(int nil, int nxl, int block) { const long c = (256l << 20) / ((long)nil * nxl * 8); return (int)std::max(1l, std::min((long)block, c)); }
}  // namespace

extern "C" {

int p3d_multi_stats(int ndev, const int* devices, int nil, int nxl, const void* x, int dtype, int nslices, double* stats)
{
    if (!x || !stats) return fail(P3D_ERR_INVALID, "NULL buffer");
    if (dtype != P3D_C64 && dtype != P3D_F32) return fail(P3D_ERR_INVALID, "unknown dtype %d", dtype);
    const size_t per = (size_t)nil * nxl * (dtype == P3D_C64 ? sizeof(c32) : sizeof(float));
    return run_blocks(ndev, devices, nslices, [&](BlockJob& j) -> int {
        const int step = chunk_slices(nil, nxl, j.hi - j.lo);
        p3d_plan* plan = nullptr;
        int rc = p3d_plan_create(&plan, j.dev, nil, nxl, step);
        for (int lo = j.lo; rc == P3D_OK && lo < j.hi; lo += step) {
            const int n = std::min(step, j.hi - lo);
            rc = p3d_pocs_stats(plan, (const char*)x + per * lo, dtype, n, stats + (size_t)P3D_STATS_PER_SLICE * lo);
        }
        const std::string keep = rc != P3D_OK ? std::string(p3d_last_error()) : std::string();
        if (plan) p3d_plan_destroy(plan);
        if (rc != P3D_OK) g_err = keep;
        return rc;
    });
}

int p3d_multi_run(int ndev, const int* devices, int nil, int nxl, const void* x, int dtype, const float* mask, const double* tau,
                  const uint8_t* active, const p3d_pocs_params* prm, void* out, int nslices, int32_t* niter_done, double* sums)
{
    if (!x || !mask || !tau || !prm || !out) return fail(P3D_ERR_INVALID, "NULL argument");
    if (dtype != P3D_C64 && dtype != P3D_F32) return fail(P3D_ERR_INVALID, "unknown dtype %d", dtype);
    if (prm->niter < 1) return fail(P3D_ERR_INVALID, "niter = %d", prm->niter);
    const size_t per = (size_t)nil * nxl * (dtype == P3D_C64 ? sizeof(c32) : sizeof(float));
    const int niter = prm->niter;
    return run_blocks(ndev, devices, nslices, [&](BlockJob& j) -> int {
        const int step = chunk_slices(nil, nxl, j.hi - j.lo);
        p3d_plan* plan = nullptr;
        int rc = p3d_plan_create(&plan, j.dev, nil, nxl, step);
        std::vector<double> part(sums ? (size_t)(niter + 1) * step : 0);
        for (int lo = j.lo; rc == P3D_OK && lo < j.hi; lo += step) {
            const int n = std::min(step, j.hi - lo);
            rc = p3d_pocs_run(plan, (const char*)x + per * lo, dtype, mask, tau + (size_t)2 * niter * lo, active ? active + lo : nullptr, prm,
                              (char*)out + per * lo, n, niter_done ? niter_done + lo : nullptr, sums ? part.data() : nullptr, nullptr);
            if (rc == P3D_OK && sums)   // [(niter + 1)][n] of this chunk -> columns lo .. lo + n of [(niter + 1)][nslices]
                for (int k = 0; k <= niter; ++k) std::memcpy(sums + (size_t)k * nslices + lo, part.data() + (size_t)k * n, sizeof(double) * n);
        }
        const std::string keep = rc != P3D_OK ? std::string(p3d_last_error()) : std::string();
        if (plan) p3d_plan_destroy(plan);
        if (rc != P3D_OK) g_err = keep;
        return rc;
    });
}

}  // extern "C"
