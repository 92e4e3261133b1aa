// Host side of libddn: plan management, transform factorisation, launch sequencing of the four
// solver loops, and the extern "C" entry points declared in include/ddn.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/ddn.h"
#include "ddn_kernels.hpp"
#include "ddn_launch.hpp"
#include "ddn_long_args.hpp"

#define DDN_DECL_ROW(T)                                                                                              \
    extern template int row_launch<T, PdrP<T>>(hipStream_t, int, int, size_t, int, const RowCommon<T>&, const PdrP<T>::Args&);   \
    extern template int row_launch<T, PddrP<T>>(hipStream_t, int, int, size_t, int, const RowCommon<T>&, const PddrP<T>::Args&); \
    extern template int row_launch<T, AdmmP<T>>(hipStream_t, int, int, size_t, int, const RowCommon<T>&, const AdmmP<T>::Args&); \
    extern template int cp_launch<T>(hipStream_t, int, int, size_t, int, const RowCommon<T>&, const CpArgs<T>&);
DDN_DECL_ROW(float)
DDN_DECL_ROW(double)
#define DDN_DECL_LONG(T)                                                                                                         \
    extern template int long_row_launch<T, PdrP<T>>(hipStream_t, int, const LongArgs<T>&, const PdrP<T>::Args&);                 \
    extern template int long_row_launch<T, PddrP<T>>(hipStream_t, int, const LongArgs<T>&, const PddrP<T>::Args&);               \
    extern template int long_row_launch<T, AdmmP<T>>(hipStream_t, int, const LongArgs<T>&, const AdmmP<T>::Args&);               \
    extern template int long_cp_launch<T>(hipStream_t, int, const LongArgs<T>&, const CpArgs<T>&);                                \
    extern template int long_plain_launch<T>(hipStream_t, int, const LongArgs<T>&, const T*, T*, T);
DDN_DECL_LONG(float)
DDN_DECL_LONG(double)
extern template int whole_row_launch<PdrP<float>>(hipStream_t, int, const WholeRowArgs&, const PdrP<float>::Args&);
extern template int whole_row_launch<PddrP<float>>(hipStream_t, int, const WholeRowArgs&, const PddrP<float>::Args&);
extern template int whole_row_launch<AdmmP<float>>(hipStream_t, int, const WholeRowArgs&, const AdmmP<float>::Args&);

namespace {


// ------------------------------------------------------------------------------ factorisation
constexpr int kMaxPrimeRadix = 8191;   // largest prime below 8 x 1024 threads (see fft_min_threads)

bool factorize(int n, FftDesc& d) {
    std::memset(&d, 0, sizeof(d));
    d.n = n;
    std::vector<int> rad;
    int rem = n;
    int twos = 0;
    while (rem % 2 == 0) {
        rem /= 2;
        ++twos;
    }
    // 2^twos as radix 16 / 8 / 4 / 2 passes, larger radices first
    if (twos > 0) {
        int a = twos / 4;
        const int r = twos % 4;
        if (r == 0) {
            for (int i = 0; i < a; ++i) rad.push_back(16);
        } else if (r == 3) {
            for (int i = 0; i < a; ++i) rad.push_back(16);
            rad.push_back(8);
        } else if (r == 2) {
            if (a >= 1) {
                for (int i = 0; i < a - 1; ++i) rad.push_back(16);
                rad.push_back(8);
                rad.push_back(8);
            } else {
                rad.push_back(4);
            }
        } else {
            if (a >= 1) {
                for (int i = 0; i < a - 1; ++i) rad.push_back(16);
                rad.push_back(8);
                rad.push_back(4);
            } else {
                rad.push_back(2);
            }
        }
    }
    const int primes[] = {3, 5, 7, 11, 13};
    for (int p : primes) {
        while (rem % p == 0) {
            rad.push_back(p);
            rem /= p;
        }
    }
    // larger prime factors (17 .. kMaxPrimeRadix): the runtime-radix pass of ddn_fft.hpp (O(p) work per element; a fallback that
    // makes such sizes run, not a fast path)
    for (int q = 17; q <= kMaxPrimeRadix && rem > 1; q += 2) {
        while (rem % q == 0) {
            rad.push_back(q);
            rem /= q;
        }
    }
    if (rem != 1) return false;
    if (rad.empty()) rad.push_back(1);
    if ((int)rad.size() > DDN_MAXPASS) return false;
    if (n == 1) {
        d.npass = 0;
        d.pow2 = 1;
        return true;
    }
    d.npass = (int)rad.size();
    d.pow2 = 1;
    int L = n;
    for (int s = 0; s < d.npass; ++s) {
        const int r = rad[s];
        d.radix[s] = r;
        d.m[s] = L / r;
        d.twmul[s] = n / L;
        int ms = -1;
        if ((d.m[s] & (d.m[s] - 1)) == 0) {
            ms = 0;
            while ((1 << ms) < d.m[s]) ++ms;
        }
        d.mshift[s] = ms;
        auto magic = [](int dv) { return dv <= 1 ? 0u : (unsigned)((((unsigned long long)1 << 32) + (unsigned)dv - 1) / (unsigned)dv); };
        d.rmagic[s] = magic(r);
        d.mmagic[s] = magic(d.m[s]);
        if ((r & (r - 1)) == 0) {
            int rs = 0;
            while ((1 << rs) < r) ++rs;
            d.rshift[s] = rs;
        } else {
            d.pow2 = 0;
        }
        L /= r;
    }
    if (!d.pow2) {
        for (int s = 0; s < d.npass; ++s) d.rshift[s] = 0;
    }
    return true;
}

// The runtime-radix pass of ddn_fft.hpp (prime factors above 13) evaluates at most two outputs per thread and round
// (eight for radices above 2048), and a round is at least one whole butterfly: a workgroup that runs a pass of radix r
// needs 2 * threads >= r (8 * threads >= r).
int fft_min_threads(const FftDesc& d) {
    int m = 64;
    for (int s = 0; s < d.npass; ++s) {
        const int r = d.radix[s];
        if (r > 16) m = std::max(m, ((r + (r > 2048 ? 7 : 1)) / (r > 2048 ? 8 : 2) + 63) / 64 * 64);
    }
    return m;
}

template <typename T>
int make_table(int n, cx<T>** out) {
    // layout read by tw_setup(): [direct n][hi tw_hi_entries(n)][lo 64]
    const int nh = tw_hi_entries(n);
    std::vector<cx<T>> h((size_t)n + nh + 64);
    auto w = [&](long e) {
        const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)(e % n) / (long double)n;
        cx<T> r;
        r.x = (T)cosl(ang);
        r.y = (T)sinl(ang);
        return r;
    };
    for (int e = 0; e < n; ++e) h[e] = w(e);
    for (int k = 0; k < nh; ++k) h[n + k] = w(64L * k);
    for (int k = 0; k < 64; ++k) h[n + nh + k] = w(k);
    HIPCHK(hipMalloc((void**)out, sizeof(cx<T>) * h.size()));
    HIPCHK(hipMemcpy(*out, h.data(), sizeof(cx<T>) * h.size(), hipMemcpyHostToDevice));
    return DDN_OK;
}

struct EvPair {
    hipEvent_t a, b;
};

}  // namespace

struct ddn_plan {
    int H = 0, W = 0, Wh = 0, batch = 0, dtype = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    FftDesc fw{}, fh{}, fa{}, fb{};
    int A = 1, B = 1;          // column transform split H = A x B (A = 1: one pass)
    int cs = 0, ntile = 1;     // spectral arrays are stored in tiles of C = 1 << cs columns
    int tile_pad_rows = 0;
    size_t spec_plane = 0;     // complex elements of one image's half-spectrum (ntile * H * C)
    int col_threads = 0;
    int row_threads = 0;       // 0 = heuristic
    int tune_cols = 0, tune_A = 0;
    int num_cu = 256;
    // device buffers (typed by dtype)
    void* twW = nullptr;
    // a plan of the row-slab decomposition that holds the multipliers of its own spectral tiles only (ddn_plan_create_slab):
    // eK and G cover tiles [slab_lo, slab_lo + slab_n); no S0 / S1 / Gc / tmp_mult; only the ddn_slab_* entries run on it
    bool restricted = false;
    int slab_lo = 0, slab_n = 0;
    void* wtw = nullptr;     // per-lane twiddles of the register-resident 4096-point transform (float) or null
    void* wtwH = nullptr;    // the same table for the H = 512 / 1024 / 2048 column transforms, or null
    void* wtwW = nullptr;    // per-lane twiddles w_W^(lane k), k < W / 64, of the W = 512 / 1024 / 2048 row transforms, or null
    void* nyqd = nullptr;    // dense home of spectral column W/2 while a solver loop runs on the wave kernels
    bool nyqd_on = false;
    bool ilv_on = false;     // a solver loop runs with S0 / S1 interleaved element by element in the S0 block (wave kernels)
    void* eKc = nullptr;     // column-major natural-order copies of eK and G for the wave-per-column stage
    void* Gcm = nullptr;
    bool eKc_ok = false, Gcm_ok = false;
    // ssim as the in-loop loss (ddn_plan_set_ssim): window taps, the caller's record, plan-owned scratch
    float ssim_g[11] = {};
    double ssim_max = 1.0;
    double* ssim_ring = nullptr;   // caller-owned [iter][image][2]; null = off
    int ssim_b_every = 50;         // primal DR: ssim(z1, b) is recorded at iterations 1, 1 + n, 1 + 2 n, ... (the reference's logging cadence)
    double* ssim_part = nullptr;   // per-tile partial sums
    double* ssim_val = nullptr;    // [batch]
    void* ssim_prev = nullptr;     // copy of the previous iterate (primal DR compares z1 with z1prev)
    // rows wider than one workgroup's LDS holds (ddn_long.hpp): W = long_R x long_M
    int long_R = 0, long_M = 0;
    FftDesc fl{};              // the length-long_M transform
    void* twL = nullptr;       // its twiddle table
    void* twHalf = nullptr;    // twiddle table of length W / 2 for the whole-row kernels (W = 8192 / 16384 / 32768, float), or null
    void* edL = nullptr;       // analytic row-spectral D, eD[k] = 1 - w_W^k, k <= W/2
    void* Ylong = nullptr;     // [Ylong_narr][rows][R/2 + 1][M] complex, sized for Ylong_rows rows
    size_t Ylong_rows = 0;
    // path switches for A/B measurements and tests, read from the environment ONCE, at plan creation (a launch costs no
    // getenv, and a plan keeps the paths it was built for even if the process changes the environment afterwards)
    struct Env {
        bool no_wave_cols = false, no_wave_rows = false, force_wave_rows = false, no_whole_rows = false, no_ilv = false,
             no_nyq_dense = false, no_nyq_fir = false, no_pdr_twin = false, no_taps = false;
        int debug_mode = 0;
        void read() {
            auto on = [](const char* n) { return getenv(n) != nullptr; };
            no_wave_cols = on("DDN_NO_WAVE_COLS"); no_wave_rows = on("DDN_NO_WAVE_ROWS"); force_wave_rows = on("DDN_FORCE_WAVE_ROWS");
            no_whole_rows = on("DDN_NO_WHOLE_ROWS"); no_ilv = on("DDN_NO_ILV"); no_nyq_dense = on("DDN_NO_NYQ_DENSE");
            no_nyq_fir = on("DDN_NO_NYQ_FIR"); no_pdr_twin = on("DDN_NO_PDR_TWIN"); no_taps = on("DDN_NO_TAPS");
            debug_mode = getenv("DDN_DEBUG_MODE") ? atoi(getenv("DDN_DEBUG_MODE")) : 0;
        }
    } env;
    int Ylong_narr = 0;        // arrays it holds (1: Chambolle-Pock and plain transforms, 3: the resolvent solvers)
    void* nyq_taps = nullptr;  // blur taps transformed along the rows at column W/2 (real, kh entries): see RowCommon::nyq_taps
    int nyq_kh = 0, nyq_ph = 0;
    bool nyq_fir_on = false;   // a Chambolle-Pock loop runs with column W/2 handled by the row kernels
    void* ctab = nullptr;    // [Wh][2][32] complex: the blur taps transformed along the rows, for the H = 4096 column stage (wcol::taps_mult)
    bool ctab_ok = false;
    double g_cK = 0, g_cD = 0;  // coefficients of the closed-form resolvent multiplier (when Gcm_ok)
    void* twH = nullptr;   // table of length H (outer twiddles and single-pass column transform)
    void* twA = nullptr;
    void* twB = nullptr;
    void* S0 = nullptr;
    void* S1 = nullptr;
    void* eK = nullptr;      // [H][Wp] complex, position order
    void* G = nullptr;       // real resolvent multiplier (includes 1/(HW))
    void* Gc = nullptr;      // complex resolvent multiplier (injected M)
    void* eDrow = nullptr;   // injected row-spectral D table (Wh entries) or null
    void* ed2 = nullptr;     // |eD|^2 table (Wh entries)
    void* tmp_mult = nullptr;  // scratch multiplier [H][Wp] complex
    double* partials = nullptr;  // [batch*H][4]
    int* stop = nullptr;         // [batch] stop flags, then one int of scratch for device-side yes / no answers
    int* stop_scratch() const { return stop + batch; }
    bool have_K = false, have_G = false, injected = false;
    bool slab_cols = false;      // a column stage over a tile range of a row-slab decomposition: tile-generic kernels only
    int slab_pitch = 0;          // while slab_cols: tile pitch (rows) of the spectral buffers when it is not H
    int rb_shift = 0;            // while slab_cols: rows stored in per-source blocks of 1 << rb_shift rows,
    long long rb_extra = 0;      //   block i displaced by i * rb_extra rows (see ColArgs)
    long long* dbg = nullptr;    // diagnostic phase stamps (DDN_DEBUG_STAMPS=1), 16 per workgroup
    // timing
    bool timing = false;
    int timing_stride = 1;       // every n-th row launch and every n-th column launch is bracketed by events
    unsigned timing_count[2] = {0, 0};
    std::vector<EvPair> ev_row, ev_col;
    std::vector<EvPair> ev_pool;
    size_t ev_used = 0;
};

namespace {

template <typename T>
size_t col_lds_bytes(int n, int cshift, int narr) {
    return (size_t)narr * lds_round(n << cshift) * sizeof(cx<T>) + (size_t)tw_lds_entries(n) * sizeof(cx<T>) + 16 * 4 * sizeof(double);
}

SpecLayout layout_of(const ddn_plan* p) {
    SpecLayout l;
    l.H = p->slab_pitch ? p->slab_pitch : p->H + p->tile_pad_rows;  // row pitch of a tile (rows beyond H are padding)
    l.cs = p->cs;
    l.ntile = p->ntile;
    return l;
}

// Choose the column tiling: tile width C (also the storage tile of every spectral array) and, when
// a whole column of C-wide tiles does not fit LDS, the split H = A x B.  Returns false if nothing fits.
template <typename T>
bool choose_col(ddn_plan* p) {
    const int H = p->H;
    const int want = (sizeof(T) == 4) ? 16 : 8;     // columns for a 128-byte piece
    const int minc = (sizeof(T) == 4) ? 8 : 4;      // 64-byte piece: still acceptable for one pass
    auto fit = [&](int n, int narr) {
        int cs = -1;
        for (int c = 0; (1 << c) <= 32; ++c) {
            if (col_lds_bytes<T>(n, c, narr) <= kLdsMax - 1024) cs = c;
        }
        return cs;
    };
    int A = 1;
    if (p->tune_A > 0) {
        A = p->tune_A;
        if (H % A) return false;
    } else {
        int cs = fit(H, 2);
        // Between the heights whose columns fit the LDS eight (four in double) at a time and those that must be split
        // there is a range -- 1243 ... 2490 rows -- where four (two) columns fit: ONE pass over 32-byte pieces then beats
        // three passes of the split transform whenever the transform has at most four radix passes (measured, one
        // image, Mpx-it/s, split / one pass: 1280^2 6.7 k / 12.7 k, 1536^2 7.2 / 14.0, 1920^2 8.6 / 18.4 -- 15.7 with the
        // split's tiles 16 wide -- against 2160^2 11.7 / 11.3 and 2400^2 11.9 / 11.9 with five passes; double 1920^2
        // 10.2 / 11.5).
        // (five passes in float since the common heights have compile-time column transforms: 1440^2 14.0 k / 16.8 k,
        // 2160^2 14.5 / 15.2, 2400^2 16.1 / 16.0)
        static const int one_pass_maxpass = getenv("DDN_ONE_PASS_MAXPASS") ? atoi(getenv("DDN_ONE_PASS_MAXPASS")) : (sizeof(T) == 4 ? 5 : 4);
        const bool narrow_one_pass = cs >= 0 && (1 << cs) >= minc / 2 && p->fh.npass <= one_pass_maxpass;
        if (!narrow_one_pass && (cs < 0 || (1 << cs) < minc)) {
            // split H = A x B, A the largest factor <= sqrt(H)
            A = 1;
            for (int a = 2; (long)a * a <= (long)H; ++a) {
                if (H % a == 0) A = a;
            }
            if (A == 1 && cs < 0) return false;   // (a prime height that fits with narrower tiles keeps them: slow, but it runs)
        }
    }
    p->A = A;
    p->B = H / A;
    if (!factorize(p->A, p->fa) || !factorize(p->B, p->fb)) return false;
    int c2 = fit(p->B, 2);
    if (A > 1) c2 = std::min(c2, fit(p->A, 2));
    if (c2 < 0) return false;
    int lim = 0;
    while ((1 << lim) < want) ++lim;
    if (p->tune_cols > 0) {
        lim = 0;
        while ((1 << lim) < p->tune_cols) ++lim;
    } else {
        // Small images on the tile-generic column kernels (one workgroup per tile): with 16-column tiles a 500 x 500
        // image has 16 workgroups for 256 CUs.  While both half-spectra of an image stay cache-sized (the narrow pieces
        // the row kernels then write never reach memory as partial lines), narrow the tiles until the plan's images
        // together have a workgroup for 9/16 of the CUs -- the widest tiles that do: the row kernels pay for narrow
        // pieces (measured, Mpx-it/s by columns per tile 1 / 2 / 4 / 8 -- 500^2: 5.2 / 5.2 / 4.7 / 4.2 k; 1000^2: 14.6 /
        // 15.7 / 15.5 / 13.4; 1920 x 1080: 16.5 / 18.6 / 19.7 / 18.6; 1200^2: 15.3 / 16.2 / 16.9 / 14.8; sixteen images
        // of 1000^2, 2 / 4 / 8 / 16 columns: 28.9 / 34.2 / 34.8 / 34.8).  The tile width changes no arithmetic: a batch
        // gives the results its images give one by one.
        // (Not where a register-resident kernel family applies: those are tied to the wide tiles.)
        auto wave_len = [](int n) { return n == 512 || n == 1024 || n == 2048 || n == 4096; };
        const bool wave = sizeof(T) == 4 && (wave_len(H) || wave_len(p->W));
        const size_t spectra = 2 * (size_t)H * p->Wh * sizeof(cx<T>);
        // (nor for a split transform, whose launches have a workgroup per tile AND segment, nor below the 64-byte piece)
        if (!wave && A == 1 && (1 << c2) >= minc && spectra <= ((size_t)32 << 20)) {
            const int cus = p->num_cu > 0 ? p->num_cu : 256;
            const long batch = std::max(1, p->batch);
            while (lim > 0 && batch * ((p->Wh + (1 << lim) - 1) >> lim) < (long)cus * 9 / 16) --lim;
        }
    }
    p->cs = std::min(c2, lim);
    p->ntile = (p->Wh + (1 << p->cs) - 1) >> p->cs;
    // tiles a power of two apart would put the rows that resident workgroups stream in lockstep at the
    // same offset modulo the DRAM channel interleave: pad the tile pitch for large power-of-two heights
    p->tile_pad_rows = 0;
    if (p->H >= 1024 && (p->H & (p->H - 1)) == 0) p->tile_pad_rows = getenv("DDN_TILE_PAD_ROWS") ? atoi(getenv("DDN_TILE_PAD_ROWS")) : 0;
    p->spec_plane = ((size_t)p->ntile * (p->H + p->tile_pad_rows)) << p->cs;
    return true;
}

template <typename T>
int alloc_spec(ddn_plan* p) {
    void** bufs[] = {&p->S0, &p->eK, &p->G, &p->Gc, &p->tmp_mult};
    for (void** q : bufs) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
    }
    // S0 and S1 are the two halves of ONE allocation: the wave kernels of the resolvent solvers use the
    // same memory as one array of (S0, S1) pairs
    const size_t spec = (size_t)p->batch * p->spec_plane * sizeof(cx<T>);
    HIPCHK(hipMalloc(&p->S0, 2 * spec));
    p->S1 = (char*)p->S0 + spec;
    HIPCHK(hipMemset(p->S0, 0, 2 * spec));
    const size_t m = p->spec_plane;
    HIPCHK(hipMalloc(&p->eK, m * sizeof(cx<T>)));
    HIPCHK(hipMalloc(&p->G, m * sizeof(T)));
    HIPCHK(hipMalloc(&p->Gc, m * sizeof(cx<T>)));
    HIPCHK(hipMalloc(&p->tmp_mult, m * sizeof(cx<T>)));
    HIPCHK(hipMemset(p->eK, 0, m * sizeof(cx<T>)));
    HIPCHK(hipMemset(p->G, 0, m * sizeof(T)));
    HIPCHK(hipMemset(p->Gc, 0, m * sizeof(cx<T>)));
    HIPCHK(hipMemset(p->tmp_mult, 0, m * sizeof(cx<T>)));
    return DDN_OK;
}

template <typename T>
int alloc_col_tables(ddn_plan* p) {
    if (p->twA) (void)hipFree(p->twA);
    if (p->twB) (void)hipFree(p->twB);
    p->twA = p->twB = nullptr;
    if (p->A > 1) {
        int rc;
        if ((rc = make_table<T>(p->A, (cx<T>**)&p->twA))) return rc;
        if ((rc = make_table<T>(p->B, (cx<T>**)&p->twB))) return rc;
    }
    return DDN_OK;
}

template <typename T>
int plan_alloc(ddn_plan* p) {
    int rc;
    if ((rc = make_table<T>(p->W, (cx<T>**)&p->twW))) return rc;
    if ((rc = make_table<T>(p->H, (cx<T>**)&p->twH))) return rc;
    if (sizeof(T) == 4 && (p->W == 4096 || p->H == 4096 || p->long_M == 4096)) {
        // lane l: w^(l c), c = 0..7, then w^(8 l a), a = 0..7, w = exp(-2 pi i / 4096)
        std::vector<cx<float>> tab(64 * 16);
        const long double tp = -2.0L * 3.14159265358979323846264338327950288L / 4096.0L;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 8; ++i) {
                tab[l * 16 + i] = {(float)cosl(tp * (l * i)), (float)sinl(tp * (l * i))};
                tab[l * 16 + 8 + i] = {(float)cosl(tp * (8 * l * i)), (float)sinl(tp * (8 * l * i))};
            }
        HIPCHK(hipMalloc(&p->wtw, tab.size() * sizeof(cx<float>)));
        HIPCHK(hipMemcpy(p->wtw, tab.data(), tab.size() * sizeof(cx<float>), hipMemcpyHostToDevice));
    }
    for (int side = 0; side < 2; ++side) {
        const int N = side ? p->H : p->W;
        if (sizeof(T) != 4 || !(N == 512 || N == 1024 || N == 2048)) continue;
        const int R = N / 64;
        std::vector<cx<float>> tab(64 * R);
        const long double tp = -2.0L * 3.14159265358979323846264338327950288L / (long double)N;
        for (int l = 0; l < 64; ++l)
            for (int k = 0; k < R; ++k) tab[l * R + k] = {(float)cosl(tp * (l * k)), (float)sinl(tp * (l * k))};
        void** dst = side ? &p->wtwH : &p->wtwW;
        HIPCHK(hipMalloc(dst, tab.size() * sizeof(cx<float>)));
        HIPCHK(hipMemcpy(*dst, tab.data(), tab.size() * sizeof(cx<float>), hipMemcpyHostToDevice));
    }
    if (sizeof(T) == 4 && whole_row_width(p->W) && (rc = make_table<T>(p->W / 2, (cx<T>**)&p->twHalf))) return rc;
    if (p->long_R) {
        if ((rc = make_table<T>(p->long_M, (cx<T>**)&p->twL))) return rc;
        std::vector<cx<T>> ed((size_t)p->Wh + 64);
        for (int k = 0; k < p->Wh; ++k) {
            const long double ang = -2.0L * 3.14159265358979323846264338327950288L * k / p->W;
            ed[k].x = (T)(1.0L - cosl(ang));
            ed[k].y = (T)(-sinl(ang));
        }
        HIPCHK(hipMalloc(&p->edL, ed.size() * sizeof(cx<T>)));
        HIPCHK(hipMemcpy(p->edL, ed.data(), ed.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
    }
    if ((rc = alloc_col_tables<T>(p))) return rc;
    if (!p->restricted && (rc = alloc_spec<T>(p))) return rc;
    const int Wt = p->Wh + 64;
    HIPCHK(hipMalloc(&p->eDrow, (size_t)Wt * sizeof(cx<T>)));
    HIPCHK(hipMalloc(&p->ed2, (size_t)Wt * sizeof(T)));
    std::vector<T> ed2(Wt, (T)0);
    for (int k = 0; k < p->Wh; ++k) ed2[k] = (T)(2.0L - 2.0L * cosl(2.0L * 3.14159265358979323846264338327950288L * k / p->W));
    HIPCHK(hipMemcpy(p->ed2, ed2.data(), (size_t)Wt * sizeof(T), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&p->partials, (size_t)p->batch * p->H * 4 * sizeof(double)));
    HIPCHK(hipMemset(p->partials, 0, (size_t)p->batch * p->H * 4 * sizeof(double)));
    if (getenv("DDN_DEBUG_STAMPS")) {
        HIPCHK(hipMalloc((void**)&p->dbg, 2 * 4096 * 16 * sizeof(long long)));
        HIPCHK(hipMemset(p->dbg, 0, 2 * 4096 * 16 * sizeof(long long)));
    }
    HIPCHK(hipMalloc((void**)&p->stop, ((size_t)p->batch + 1) * sizeof(int)));
    HIPCHK(hipMemset(p->stop, 0xff, (size_t)p->batch * sizeof(int)));
    return DDN_OK;
}

// ------------------------------------------------------------------------------ launch helpers
struct TimerScope {
    ddn_plan* p;
    std::vector<EvPair>* vec;
    EvPair ev{};
    bool on;
    // full = false: a launch that does only part of the steady-state work (the produce-only first and
    // the consume-only last row launch of a loop); never sampled, so the mean is that of full launches
    TimerScope(ddn_plan* p_, bool row, bool full = true) : p(p_), vec(row ? &p_->ev_row : &p_->ev_col), on(p_->timing && full) {
        if (on) on = (p->timing_count[row ? 0 : 1]++ % (unsigned)p->timing_stride) == 0;
        if (!on) return;
        if (p->ev_used == p->ev_pool.size()) {
            EvPair e;
            if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) {
                on = false;
                return;
            }
            p->ev_pool.push_back(e);
        }
        ev = p->ev_pool[p->ev_used++];
        (void)hipEventRecord(ev.a, p->stream);
    }
    ~TimerScope() {
        if (!on) return;
        (void)hipEventRecord(ev.b, p->stream);
        vec->push_back(ev);
    }
};

template <typename T>
ColMap colmap(const ddn_plan* p) {
    ColMap m;
    m.fa = p->fa;
    m.fb = p->fb;
    m.A = p->A;
    m.B = p->B;
    return m;
}

inline LossFin no_loss() {
    LossFin lf;
    std::memset(&lf, 0, sizeof(lf));
    return lf;
}

template <typename T, int MODE, int NF, bool SMALL = false>
int launch_col_fn(ddn_plan* p, ColArgs<T>& a, int nblocks, int threads, size_t lds) {
    auto fn = k_col<T, MODE, NF, SMALL>;
    int rc = ensure_lds_attr((const void*)fn);
    if (rc) return rc;
    TimerScope ts(p, false);
    hipLaunchKernelGGL(fn, dim3(nblocks), dim3(threads), lds, p->stream, a);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}
template <typename T, int MODE>
int launch_col_one(ddn_plan* p, ColArgs<T>& a, int nblocks, int threads, size_t lds) {
    // straight-line transforms instead of the run-time radix code for the lengths the large configurations use: the
    // factors of a 32768-row column (128 x 256) and of a 4096-row column that is split (64 x 64: float64, slab plans);
    // 512 rows in float64 (the 512 x 512 parity configuration)
    static const bool generic_only = getenv("DDN_NO_FIXED_COLS") != nullptr;
    static const bool no_small = getenv("DDN_NO_SMALL_COLS") != nullptr;
    // small tiles (at most four elements per thread and array) of a height without a compile-time transform: the variant
    // that overlaps its global loads (one image, 100^2 ... 700^2: +3-5 %; nothing for the compile-time heights)
    const bool small = !no_small && threads <= 512 && (a.fd.n << a.lay.cs) <= 4 * threads;
    if constexpr (sizeof(T) == 4 && (MODE == COL_RES2 || MODE == COL_MUL)) {
        if (small && (generic_only || !(MODE == COL_RES2 ? col_fixed_res2_f32(a.fd.n) : col_fixed_mul_f32(a.fd.n))) && a.fd.n != 64 && a.fd.n != 128 && a.fd.n != 256)
            return launch_col_fn<T, MODE, 0, true>(p, a, nblocks, threads, lds);
    }
    if (!generic_only) {
        if (a.fd.n == 64) return launch_col_fn<T, MODE, 64>(p, a, nblocks, threads, lds);
        if constexpr (sizeof(T) == 4 && (MODE == COL_RES2 || MODE == COL_MUL || MODE == COL_FWD2 || MODE == COL_FWD)) {
            // common image heights and their factors: kernels of their own units
            const void* fn = MODE == COL_RES2 ? col_fixed_res2_f32(a.fd.n) : MODE == COL_MUL ? col_fixed_mul_f32(a.fd.n)
                           : MODE == COL_FWD2 ? col_fixed_fwd2_f32(a.fd.n) : col_fixed_fwd_f32(a.fd.n);
            if (fn) {
                int rc = ensure_lds_attr(fn);
                if (rc) return rc;
                TimerScope ts(p, false);
                void* args[] = {(void*)&a};
                HIPCHK(hipLaunchKernel(fn, dim3(nblocks), dim3(threads), args, lds, p->stream));
                return DDN_OK;
            }
        }
        if constexpr (sizeof(T) == 4) {
            if (a.fd.n == 256) return launch_col_fn<T, MODE, 256>(p, a, nblocks, threads, lds);
            if (a.fd.n == 128) return launch_col_fn<T, MODE, 128>(p, a, nblocks, threads, lds);
        } else {
            if (a.fd.n == 512) return launch_col_fn<T, MODE, 512>(p, a, nblocks, threads, lds);
        }
    }
    return launch_col_fn<T, MODE, 0>(p, a, nblocks, threads, lds);
}

// Threads of a tile-generic column workgroup: sixteen elements per thread, or -- while the launch's workgroups leave
// wavefronts free on the chip (a small image: one workgroup per narrow tile) -- four or eight per thread: a 500-row column pair on
// one wavefront is a chain of twenty dependent butterflies per pass (one image, column launch in us, 64 / 256 threads:
// 300^2 25 / 17, 500^2 36 / 18; 1000^2, 256 / 512 threads: 42 / 26).
inline int col_threads_for(const ddn_plan* p, const FftDesc& fd, int cs, int narr, int nblocks = 1 << 30) {
    const int n = fd.n;
    const int elems = (n << cs) * narr;
    int threads = std::min(1024, std::max(64, (elems / 16 + 63) / 64 * 64));
    for (int per_thread : {4, 8}) {
        const int wide = std::min(1024, std::max(64, (elems / per_thread + 63) / 64 * 64));
        if (wide > threads && (long)nblocks * (wide / 64) <= (long)p->num_cu * 8) {
            threads = wide;
            break;
        }
    }
    if (p->col_threads > 0) threads = std::max(64, std::min(1024, p->col_threads / 64 * 64));
    return std::max(threads, fft_min_threads(fd));
}

inline int ew_grid(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 256 * 8); }

constexpr double kColStaggerUs = 8.0;   // default delay of the late half of the column stage's first round
inline bool wcol_ready(const ddn_plan* p) {
    return p->eKc_ok && p->wtw && p->H == 4096 && p->cs == 4 && !p->injected && !p->slab_cols && !p->env.no_wave_cols;
}
// H = 512 / 1024 / 2048: several adjacent columns per wavefront (ddn_wcoln.hpp)
inline bool wcoln_ready(const ddn_plan* p) {
    return p->eKc_ok && p->wtwH && (1 << p->cs) >= 4096 / p->H && p->cs <= 4 && !p->injected && !p->slab_cols && !p->env.no_wave_cols;
}
// column-major natural-order copy of a tiled multiplier (complex and / or real)
template <typename T>
int make_colmajor(ddn_plan* p, const cx<T>* tiled, const T* tiled_real, void** outc, void** outr) {
    const size_t total = (size_t)p->H * p->Wh;
    const size_t padded = (size_t)wcol::kColPitch * p->Wh;
    if (outc && !*outc) HIPCHK(hipMalloc(outc, padded * sizeof(cx<T>)));
    if (outr && !*outr) HIPCHK(hipMalloc(outr, padded * sizeof(T)));
    hipLaunchKernelGGL(wcol::k_mult_colmajor<T>, dim3(ew_grid(total)), dim3(256), 0, p->stream, tiled, tiled_real,
                       outc ? (cx<T>*)*outc : nullptr, outr ? (T*)*outr : nullptr, p->H, p->Wh, layout_of(p), colmap<T>(p));
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

// One full column stage (forward, pointwise algebra, inverse) in place on S0 (and S1).
// MODE as in ddn_kernels.hpp; nplanes = images to process.
template <typename T, int MODE>
int col_stage(ddn_plan* p, int nplanes, const cx<T>* mult, const T* greal, const cx<T>* gcplx, int conj_mult, T scale, const LossFin& lf) {
    constexpr int NARR = (MODE == COL_RES2) ? 2 : 1;
    if constexpr (sizeof(T) == 4 && (MODE == COL_RES2 || MODE == COL_MUL)) {
        // H = 4096: one pass, one wavefront per column (ddn_wcol.hpp)
        const bool wave1 = wcol_ready(p), waven = !wave1 && wcoln_ready(p);
        if ((wave1 || waven) && mult == (const cx<T>*)p->eK && !gcplx && (MODE == COL_MUL || (greal == (const T*)p->G && p->Gcm_ok))) {
            wcol::Args w;
            std::memset(&w, 0, sizeof(w));
            w.S0 = (cx<float>*)p->S0; w.S1 = (cx<float>*)p->S1;
            w.eKc = (const cx<float>*)p->eKc; w.wtw = (const cx<float>*)p->wtw; w.twH = (const cx<float>*)p->twH;
            w.ed2 = (const float*)p->ed2; w.cK = (float)p->g_cK; w.cD = (float)p->g_cD; w.gscale = (float)(1.0 / ((double)p->H * (double)p->W));
            w.lay = layout_of(p); w.Wh = p->Wh; w.batch = nplanes;
            const bool fir = MODE == COL_MUL && p->nyq_fir_on;   // column W/2 is the row kernels' business
            if (fir) w.Wh = p->Wh - 1;
            w.ctab = (wave1 && p->ctab_ok) ? (const cx<float>*)p->ctab : nullptr;
            w.conj_mult = conj_mult; w.scale = (float)scale; w.lf = lf;
            w.nyqd = (p->nyqd_on && !fir) ? (float*)p->nyqd : nullptr;
            w.ilv = (p->ilv_on && MODE == COL_RES2) ? 1 : 0;
            w.dbg = p->dbg ? p->dbg + 4096 * 16 : nullptr;  // second half of the stamp buffer
            if (wave1 && MODE == COL_RES2) {
                // half of the first round starts late (k_wave_col): DDN_COL_STAGGER_US overrides the delay, 0 switches it off
                static const double stagger_us = getenv("DDN_COL_STAGGER_US") ? atof(getenv("DDN_COL_STAGGER_US")) : kColStaggerUs;
                w.stagger = (int)(std::min(50.0, std::max(0.0, stagger_us)) * 100.0);   // ticks of the 100 MHz wall clock; at most 50 us
                w.stagger_blocks = p->num_cu;     // the first round: one eight-wave workgroup per compute unit
            }
            TimerScope ts(p, false);
            if (waven) {
                w.wtw = (const cx<float>*)p->wtwH;
                return wcoln_launch(p->stream, MODE, w);
            }
            return wcol_launch(p->stream, MODE, w);
        }
    }
    ColArgs<T> a;
    std::memset(&a, 0, sizeof(a));
    a.S0 = (cx<T>*)p->S0;
    a.S1 = (cx<T>*)p->S1;
    a.mult = mult;
    a.greal = greal;
    a.gcplx = gcplx;
    a.twH = (const cx<T>*)p->twH;
    a.lay = layout_of(p);
    a.batch = nplanes;
    a.conj_mult = conj_mult;
    a.scale = scale;
    a.lf = no_loss();
    if (p->slab_pitch) {
        a.mult_pitch = p->H + p->tile_pad_rows;
        a.rb_shift = p->rb_shift;
        a.rb_extra = p->rb_extra;
    }
    const int cs = p->cs;
    if (p->A == 1) {
        a.fd = p->fb;
        a.tw = (const cx<T>*)p->twH;
        a.nseg = 1; a.row_stride = 1; a.seg_stride = 0;
        a.outer = 0;
        a.lf = lf;
        return launch_col_one<T, MODE>(p, a, nplanes * a.lay.ntile, col_threads_for(p, a.fd, cs, NARR, nplanes * a.lay.ntile), col_lds_bytes<T>(a.fd.n, cs, NARR));
    }
    // split transform: strided length-A stage, middle length-B stage (with the pointwise algebra),
    // strided length-A inverse stage
    ColArgs<T> s = a;
    s.fd = p->fa;
    s.tw = (const cx<T>*)p->twA;
    s.nseg = p->B; s.row_stride = p->B; s.seg_stride = 1;
    s.outer = 0;
    s.scale = (T)1;
    const int thr_s = col_threads_for(p, p->fa, cs, NARR);
    const size_t lds_s = col_lds_bytes<T>(p->A, cs, NARR);
    const int nb_s = nplanes * s.nseg * s.lay.ntile;
    int rc;
    s.dir = 0;
    s.lf = lf;  // the loss of the previous iteration is finalised by the first launch
    if (NARR == 2) rc = launch_col_one<T, COL_FWD2>(p, s, nb_s, thr_s, lds_s);
    else rc = launch_col_one<T, COL_FWD>(p, s, nb_s, thr_s, lds_s);
    if (rc) return rc;
    s.lf = no_loss();
    s.lf.stop = lf.stop; s.lf.iter = lf.iter;
    // middle stage
    a.fd = p->fb;
    a.fa = p->fa;
    a.tw = (const cx<T>*)p->twB;
    a.nseg = p->A; a.row_stride = 1; a.seg_stride = p->B;
    a.outer = (MODE == COL_FWD) ? 1 : 3;
    a.lf.stop = lf.stop; a.lf.iter = lf.iter;
    if ((rc = launch_col_one<T, MODE>(p, a, nplanes * a.nseg * a.lay.ntile, col_threads_for(p, a.fd, cs, NARR), col_lds_bytes<T>(a.fd.n, cs, NARR)))) return rc;
    if (MODE == COL_FWD) return DDN_OK;
    // inverse short stage
    s.dir = 1;
    if (NARR == 2) rc = launch_col_one<T, COL_FWD2>(p, s, nb_s, thr_s, lds_s);
    else rc = launch_col_one<T, COL_FWD>(p, s, nb_s, thr_s, lds_s);
    return rc;
}

struct RowCfg {
    int V, threads, grid;
    bool dense = false;     // V == 4, float, compile-time width: the variant for several rounds of workgroups per CU (row_launch_w)
    size_t lds;
    bool aligned = false;   // every array of the launch is 16-byte aligned (noted for the split-row pixel kernels)
};

// Row workgroup shape: V = 4 pixels per lane-access (16 B) and four groups per thread when the
// arrays are 16-byte aligned and W <= 4096; otherwise scalar accesses, eight pixels per thread.
template <typename T>
bool row_cfg(const ddn_plan* p, bool aligned, int work_items, RowCfg& rc, bool cp = false, bool admm = false) {
    const int W = p->W;
    if (p->long_R) {
        // rows wider than 8192 pixels: the three-kernel split transform (ddn_long.hpp); V = -1 marks it
        rc.V = -1; rc.threads = 256; rc.lds = 0; rc.grid = 1; rc.aligned = aligned;
        return true;
    }
    // A wavefront of the register-resident kernels works through its rows alone (two transforms and the pixel
    // loop in sequence, 35-40 us whatever the image size); with few wavefronts the workgroup-per-row LDS kernels
    // finish sooner.  W = 4096: fewer than a quarter of the CUs (half for Chambolle-Pock, whose wavefronts carry twice
    // the rows).  W = 512 / 1024 / 2048, where the LDS kernels with the compile-time transform and 16-byte accesses
    // compete (measured, one to sixteen images, row launch in us, LDS / wavefront kernels -- primal DR: 512^2 x 8 41 / 44,
    // 1024^2 x 2 49 / 42, x 4 84 / 49; ADMM, whose nine arrays leave the wavefront kernel two loads in flight: 512^2 x 8
    // 46 / 80, x 16 81 / 97, 1024^2 x 4 90 / 91; Chambolle-Pock: 1024^2 x 2 28 / 38, x 4 44 / 42): fewer than two
    // wavefronts per CU, four for ADMM.  DDN_FORCE_WAVE_ROWS=1 / DDN_NO_WAVE_ROWS=1 override.
    const int wave_items = (W == 4096) ? work_items : p->batch * p->H / ((cp ? 2 : 1) * std::max(1, 4096 / W));
    int wave_thresh = cp ? p->num_cu / 2 : p->num_cu / 4;
    if (W != 4096 && aligned && W % 4 == 0) wave_thresh = admm ? 4 * p->num_cu : 2 * p->num_cu - 1;
    const bool wave_pays = wave_items > wave_thresh || p->env.force_wave_rows;
    if (sizeof(T) == 4 && p->wtwW && !p->injected && p->H % ((cp ? 2 : 1) * (4096 / W)) == 0 && wave_pays && !p->env.no_wave_rows) {
        // one wavefront per 4096 / W consecutive rows, twice as many for Chambolle-Pock (ddn_wrown.hpp)
        rc.V = 0; rc.threads = 64; rc.lds = 0;
        const int items = p->batch * p->H / ((cp ? 2 : 1) * (4096 / W));
        rc.grid = std::max(1, std::min(items, p->num_cu * 8));
        return true;
    }
    if (sizeof(T) == 4 && p->wtw && W == 4096 && p->cs == 4 && !p->injected && wave_pays && !p->env.no_wave_rows) {
        // one wavefront per row, eight per CU (256 registers each), the row held in registers
        rc.V = 0; rc.threads = 64; rc.lds = 0;
        rc.grid = std::max(1, std::min(work_items, p->num_cu * 8));
        return true;
    }
    if (sizeof(T) == 4 && p->twHalf && !p->injected && aligned && !p->env.no_whole_rows) {
        // 8192-pixel rows: the whole-row kernel (ddn_wholerow.hpp) instead of scalar accesses
        // and the run-time transform below; V = -1 is resolved in launch_solver_row
        rc.V = -1; rc.threads = 256; rc.lds = 0; rc.grid = 1; rc.aligned = aligned;
        return true;
    }
    // (float rows of 4097 ... 8192 pixels: the 16-byte-access kernel with 512 threads instead of scalar accesses -- one
    // 6000 x 4000 image 13.3 k Mpx-it/s before; not with injected eigen arrays, whose kernels are built for 256 threads)
    // (nor with a radix above 16: the run-time prime pass wants the scalar kernel's thousand threads -- 5472 = 2^5 3^2 19:
    // 7.0 k against 6.5 k)
    bool small_radices = true;
    for (int s = 0; s < p->fw.npass; ++s) small_radices = small_radices && p->fw.radix[s] <= 16;
    const int vec_max_w = (sizeof(T) == 4 && !p->injected && small_radices) ? 8192 : 4096;
    if (aligned && W % 4 == 0 && W <= vec_max_w && fft_min_threads(p->fw) <= 256) {   // (256: the launch bound of most V = 4 kernels)
        rc.V = 4;
        // float kernels specialised for these widths use eight pixels per thread (W/8 threads, radix-8
        // passes); every other V = 4 kernel sixteen pixels per thread (W/16 threads)
        // (double kernels of these widths too, with or without injected eigen arrays: row_launch_d)
        const bool pow2w = W == 512 || W == 1024 || W == 2048 || W == 4096;
        const bool eight = (sizeof(T) == 8 && pow2w) || (sizeof(T) == 4 && !p->injected && roww_fixed_width(W));
        rc.threads = eight ? (W / 8 + 63) / 64 * 64 : std::min(W > 4096 ? 512 : 256, std::max(64, ((W + 15) / 16 + 63) / 64 * 64));
        if (!eight && sizeof(T) == 4 && !p->injected) {
            // a row gets more threads (8 or 4 pixels each instead of 16) while the image's rows leave wavefronts for it
            // (eight per CU): a 1000-pixel row on one wavefront takes 74 us, and 1000 rows are half of what the chip holds
            // (one 1000^2 image: 7.1 -> 8.8 k Mpx-it/s with two wavefronts per row; 1920 rows already fill it: 18.3 -> 16.5 k)
            for (int per_thread : {4, 8}) {
                const int thr = std::max(64, ((W + per_thread - 1) / per_thread + 63) / 64 * 64);
                if (thr <= 512 && work_items <= p->num_cu * (8 / (thr / 64))) {     // (whole workgroups per CU: 8 / wavefronts each)
                    rc.threads = std::max(rc.threads, thr);
                    break;
                }
            }
        }
        static const bool wide16 = getenv("DDN_WIDE16") != nullptr;
        if (!wide16 && sizeof(T) == 4 && W > 4096 && !p->injected && roww_wide_fixed_width(W)) rc.threads = (W / 8 + 63) / 64 * 64;   // (row_launch: eight pixels per thread)
        if (!eight) rc.threads = std::max(rc.threads, fft_min_threads(p->fw));   // at most 256: within the kernels' launch bound
    } else {
        if (W > 8192) return false;
        rc.V = 1;
        rc.threads = std::max(std::min(1024, std::max(64, ((W + 7) / 8 + 63) / 64 * 64)), fft_min_threads(p->fw));
    }
    rc.lds = row_lds_bytes<T>(W, rc.V);
    if (rc.lds > kLdsMax) return false;
    int per_cu = (int)std::min<size_t>(8, kLdsMax / rc.lds);
    per_cu = std::max(1, std::min(per_cu, 2048 / rc.threads));
    // the V = 4 kernels are compiled for two wavefronts per SIMD, eight per CU: as many workgroups as that holds, so
    // that a small image has all its rows in flight
    if (rc.V == 4) per_cu = std::min(per_cu, std::max(1, 512 / rc.threads));
    rc.dense = false;
    if (rc.V == 4 && sizeof(T) == 4 && !p->injected && roww_fixed_width(W)) {
        static const double dense_k = getenv("DDN_ROW_DENSE_K") ? atof(getenv("DDN_ROW_DENSE_K")) : 1.0;
        const int dense_per_cu = (int)std::min<size_t>(std::max(1, 1024 / rc.threads), kLdsMax / rc.lds);
        if (dense_per_cu > per_cu && (double)work_items > dense_k * p->num_cu * per_cu) {
            rc.dense = true;
            per_cu = dense_per_cu;
        }
    }
    rc.grid = std::max(1, std::min(work_items, p->num_cu * per_cu));
    return true;
}

template <typename T>
RowCommon<T> row_common(ddn_plan* p, int V, bool consume, bool produce) {
    RowCommon<T> c;
    std::memset(&c, 0, sizeof(c));
    c.H = p->H; c.W = p->W; c.batch = p->batch;
    c.V_groups = V ? p->W / V : 0;
    c.wtw = (const cx<T>*)(p->W == 4096 ? p->wtw : p->wtwW);
    c.nyqd = p->nyqd_on ? (T*)p->nyqd : nullptr;
    c.ilv = p->ilv_on ? 1 : 0;
    c.fw = p->fw;
    c.lay = layout_of(p);
    c.twW = (const cx<T>*)p->twW;
    c.S0 = (cx<T>*)p->S0;
    c.S1 = (cx<T>*)p->S1;
    c.eDrow = p->injected ? (const cx<T>*)p->eDrow : nullptr;
    c.stop = p->stop;
    c.partials = p->partials;
    c.do_consume = consume ? 1 : 0;
    c.do_produce = produce ? 1 : 0;
    c.dbg = p->dbg;
    c.dbg_mode = p->env.debug_mode;
    return c;
}

inline bool aligned16(const void* q) { return ((uintptr_t)q & 15) == 0; }

// ---- long rows -------------------------------------------------------------------------------------------
template <typename T>
int long_args(ddn_plan* p, const RowCommon<T>& c, int narr, LongArgs<T>& a) {
    const size_t rows = (size_t)c.batch * c.H;
    if (rows * narr > p->Ylong_rows * p->Ylong_narr) {
        if (p->Ylong) {
            HIPCHK(hipStreamSynchronize(p->stream));
            (void)hipFree(p->Ylong);
            p->Ylong = nullptr;
        }
        HIPCHK(hipMalloc(&p->Ylong, (size_t)narr * rows * (size_t)(p->long_R / 2 + 1) * p->long_M * sizeof(cx<T>)));
        p->Ylong_rows = rows;
        p->Ylong_narr = narr;
    }
    std::memset(&a, 0, sizeof(a));
    a.c = c;
    a.Y = (cx<T>*)p->Ylong;
    a.twM = (const cx<T>*)p->twL;
    a.ed = p->injected ? (const cx<T>*)p->eDrow : (const cx<T>*)p->edL;
    a.fm = p->fl;
    a.wtw = p->env.no_wave_rows ? nullptr : (const cx<float>*)p->wtw;
    a.R = p->long_R; a.M = p->long_M;
    a.narr = narr;
    return DDN_OK;
}

// fused solver row kernels live in their own translation units (ddn_row_*.hip), one per policy
// and dtype, so that they compile in parallel; see ddn_launch.hpp.
template <typename T, class P>
int launch_solver_row(ddn_plan* p, const RowCfg& cfg, const RowCommon<T>& c, const typename P::Args& a) {
    TimerScope ts(p, true, c.do_consume && c.do_produce);
    if (cfg.V == -1) {
        if constexpr (sizeof(T) == 4) {
            // rows of 16384 / 32768 pixels with the two-tap D: the row in one workgroup's LDS (ddn_wholerow.hpp)
            if (p->twHalf && !p->injected && cfg.aligned && !p->env.no_whole_rows) {
                WholeRowArgs wa;
                wa.c = c;
                wa.twHalf = (const cx<float>*)p->twHalf;
                return whole_row_launch<P>(p->stream, p->num_cu, wa, a);
            }
        }
        if (!p->long_R) return fail(DDN_E_STATE, "row launch: no whole-row kernel and no split for this row length");
        LongArgs<T> la;
        int rc = long_args<T>(p, c, 3, la);
        if (rc) return rc;
        la.vec = cfg.aligned && p->W % 4 == 0 ? 4 : 1;
        return long_row_launch<T, P>(p->stream, p->num_cu, la, a);
    }
    return row_launch<T, P>(p->stream, cfg.grid, cfg.threads, cfg.lds, cfg.dense ? 5 : cfg.V, c, a);
}

template <typename T>
int launch_cp_row(ddn_plan* p, const RowCfg& cfg, const RowCommon<T>& c, const CpArgs<T>& a) {
    TimerScope ts(p, true, c.do_consume && c.do_produce);
    if (cfg.V == -1) {
        if constexpr (sizeof(T) == 4) {
            if (p->twHalf && !p->injected && cfg.aligned && !p->env.no_whole_rows) {
                WholeRowArgs wa;
                wa.c = c;
                wa.twHalf = (const cx<float>*)p->twHalf;
                return whole_row_cp_launch(p->stream, p->num_cu, wa, a);
            }
        }
        if (!p->long_R) return fail(DDN_E_STATE, "row launch: no whole-row kernel and no split for this row length");
        LongArgs<T> la;
        // injected eigen arrays: D z and D^T(y1 + y2) as a second spectrum in S1 (LongArgs::cp_spec), two arrays in the dual step
        const bool spec = p->injected;
        int rc = long_args<T>(p, c, spec && a.step == 0 ? 2 : 1, la);
        if (rc) return rc;
        if (spec) {
            la.cp_spec = 1;
            la.inv_add_s1 = a.step == 1 ? 1 : 0;
            la.fwd_mode = a.step == 1 ? 1 : 2;
            la.wtw = nullptr;        // the frequency-side kernels with these modes are the generic ones
        }
        return long_cp_launch<T>(p->stream, p->num_cu, la, a);
    }
    return cp_launch<T>(p->stream, cfg.grid, cfg.threads, cfg.lds, cfg.dense ? 5 : cfg.V, c, a);
}

template <typename T>
int launch_finalize(ddn_plan* p, const LossFin& lf) {
    hipLaunchKernelGGL(k_loss_finalize, dim3(p->batch), dim3(256), 0, p->stream, lf);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

LossFin make_lf(ddn_plan* p, double* ring, int iter, int tol_kind, double tol, int first_iter, bool use_stop) {
    LossFin lf;
    std::memset(&lf, 0, sizeof(lf));
    lf.partials = p->partials;
    lf.ring = (ring && iter >= 1) ? ring + (size_t)(iter - 1) * p->batch * 8 : nullptr;
    lf.stop = use_stop ? p->stop : nullptr;
    lf.rows_per_plane = p->H;
    lf.tol_kind = tol_kind;
    lf.iter = iter;
    lf.first_iter = first_iter;
    lf.tol = tol;
    lf.npix = (double)p->H * (double)p->W;
    lf.enabled = (iter >= 1 && (ring != nullptr || tol_kind > 0)) ? 1 : 0;
    return lf;
}

template <typename T>
int reset_stop(ddn_plan* p) {
    HIPCHK(hipMemsetAsync(p->stop, 0xff, (size_t)p->batch * sizeof(int), p->stream));
    return DDN_OK;
}

template <typename T>
int finish_iters(ddn_plan* p, int* iters_done_dev, int n_iters) {
    if (!iters_done_dev) return DDN_OK;
    hipLaunchKernelGGL(k_iters_done, dim3((p->batch + 255) / 256), dim3(256), 0, p->stream, p->stop, iters_done_dev, p->batch, n_iters);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

// ------------------------------------------------------------------------- plain transforms
template <typename T>
int rows_r2c(ddn_plan* p, const T* src, int nplanes) {
    PlainRowArgs<T> a;
    std::memset(&a, 0, sizeof(a));
    a.H = p->H; a.W = p->W; a.batch = nplanes;
    a.fw = p->fw; a.lay = layout_of(p); a.twW = (const cx<T>*)p->twW; a.S = (cx<T>*)p->S0; a.src = src;
    if (p->long_R) {
        RowCommon<T> c;
        std::memset(&c, 0, sizeof(c));
        c.H = p->H; c.W = p->W; c.batch = nplanes; c.lay = layout_of(p); c.S0 = (cx<T>*)p->S0; c.S1 = (cx<T>*)p->S1; c.do_produce = 1;
        LongArgs<T> la;
        int rcl = long_args<T>(p, c, 1, la);
        if (rcl) return rcl;
        return long_plain_launch<T>(p->stream, p->num_cu, la, src, (T*)nullptr, (T)1);
    }
    const size_t lds = row_lds_bytes<T>(p->W, 1);
    if (lds > kLdsMax) return fail(DDN_E_UNSUPPORTED, "row length too large for LDS");
    auto fn = k_rows_r2c<T>;
    int rc;
    if ((rc = ensure_lds_attr((const void*)fn))) return rc;
    const int threads = std::max(std::min(1024, std::max(64, (p->W / 8 + 63) / 64 * 64)), fft_min_threads(p->fw));
    const int pairs = nplanes * ((p->H + 1) / 2);
    hipLaunchKernelGGL(fn, dim3(std::min(pairs, p->num_cu * 4)), dim3(threads), lds, p->stream, a);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}
template <typename T>
int rows_c2r(ddn_plan* p, T* dst, int nplanes, T scale, const T* add) {
    PlainRowArgs<T> a;
    std::memset(&a, 0, sizeof(a));
    a.H = p->H; a.W = p->W; a.batch = nplanes;
    a.fw = p->fw; a.lay = layout_of(p); a.twW = (const cx<T>*)p->twW; a.S = (cx<T>*)p->S0; a.dst = dst; a.scale = scale;
    (void)add;
    if (p->long_R) {
        RowCommon<T> c;
        std::memset(&c, 0, sizeof(c));
        c.H = p->H; c.W = p->W; c.batch = nplanes; c.lay = layout_of(p); c.S0 = (cx<T>*)p->S0; c.S1 = (cx<T>*)p->S1; c.do_consume = 1;
        LongArgs<T> la;
        int rcl = long_args<T>(p, c, 1, la);
        if (rcl) return rcl;
        return long_plain_launch<T>(p->stream, p->num_cu, la, (const T*)nullptr, dst, scale);
    }
    const size_t lds = row_lds_bytes<T>(p->W, 1);
    auto fn = k_rows_c2r<T>;
    int rc;
    if ((rc = ensure_lds_attr((const void*)fn))) return rc;
    const int threads = std::max(std::min(1024, std::max(64, (p->W / 8 + 63) / 64 * 64)), fft_min_threads(p->fw));
    const int pairs = nplanes * ((p->H + 1) / 2);
    hipLaunchKernelGGL(fn, dim3(std::min(pairs, p->num_cu * 4)), dim3(threads), lds, p->stream, a);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}


// y = real(ifft2(m * fft2(x))) with a position-ordered half-plane multiplier
template <typename T>
int apply_mult(ddn_plan* p, const cx<T>* mult, int conj_mult, const T* x, T* y, int nplanes) {
    int rc;
    if ((rc = rows_r2c<T>(p, x, nplanes))) return rc;
    const T scale = (T)(1.0 / ((double)p->H * (double)p->W));
    if ((rc = col_stage<T, COL_MUL>(p, nplanes, mult, nullptr, nullptr, conj_mult, scale, no_loss()))) return rc;
    return rows_c2r<T>(p, y, nplanes, (T)1, nullptr);
}

template <typename T>
int import_full(ddn_plan* p, const void* full, int cdtype, int recip, int conj_in, double scale, cx<T>* out, T* out_real) {
    const size_t total = (size_t)p->H * p->Wh;
    if (cdtype == DDN_C64)
        hipLaunchKernelGGL((k_import_mult<T, float>), dim3(ew_grid(total)), dim3(256), 0, p->stream, (const cx<float>*)full, out, out_real,
                           p->H, p->W, p->Wh, layout_of(p), colmap<T>(p), recip, conj_in, scale);
    else
        hipLaunchKernelGGL((k_import_mult<T, double>), dim3(ew_grid(total)), dim3(256), 0, p->stream, (const cx<double>*)full, out, out_real,
                           p->H, p->W, p->Wh, layout_of(p), colmap<T>(p), recip, conj_in, scale);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

// --------------------------------------------------------------------------------- set-up
template <typename T>
int set_kernel_impl(ddn_plan* p, const double* taps, int kh, int kw, bool as_convolution, cx<T>* out) {
    // impulse response image: correlation centred at (kh/2, kw/2) (eig:41-56), or the circular
    // convolution kernel of circular_convolve2d: zero-padded taps rolled by (-kh)//2 (conv:38-40)
    const int H = p->H, W = p->W;
    std::vector<T> img((size_t)H * W, (T)0);
    for (int u = 0; u < kh; ++u)
        for (int v = 0; v < kw; ++v) {
            const T tap = (T)(float)taps[u * kw + v];
            long r, c;
            if (as_convolution) {
                const long sh = -((kh + 1) / 2), sw = -((kw + 1) / 2);  // python floor division of -k by 2
                r = ((u + sh) % H + H) % H;
                c = ((v + sw) % W + W) % W;
                if (u >= H || v >= W) continue;  // the reference writes padded[:kh,:kw]; taps beyond the image would raise
            } else {
                r = (((long)kh / 2 - u) % H + H) % H;
                c = (((long)kw / 2 - v) % W + W) % W;
            }
            img[(size_t)r * W + c] += tap;
        }
    T* dimg = nullptr;
    HIPCHK(hipMalloc((void**)&dimg, sizeof(T) * (size_t)H * W));
    HIPCHK(hipMemcpyAsync(dimg, img.data(), sizeof(T) * (size_t)H * W, hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    int rc = rows_r2c<T>(p, dimg, 1);
    if (!rc) rc = col_stage<T, COL_FWD>(p, 1, nullptr, nullptr, nullptr, 0, (T)1, no_loss());
    if (!rc) {
        hipError_t e = hipMemcpyAsync(out, p->S0, sizeof(cx<T>) * p->spec_plane, hipMemcpyDeviceToDevice, p->stream);
        if (e != hipSuccess) rc = fail(DDN_E_HIP, hipGetErrorString(e));
    }
    (void)hipStreamSynchronize(p->stream);
    (void)hipFree(dimg);
    return rc;
}

// eK of the plan's own tile range straight from the taps (a plan of ddn_plan_create_slab holds no whole plane to
// transform).  The impulse response of set_kernel_impl is separable in position: tap (u, v) sits at row rpos[u],
// column cpos[v], so  eK[k1][k2] = sum_u w_H^(k1 rpos[u]) * sum_v tap[u][v] w_W^(k2 cpos[v]):  k_taps_cols builds
// the inner sums for the local columns, k_taps_eK the outer one; both in double, rounded once.
__global__ void k_taps_cols(const double* taps, const int* cpos, int kh, int kw, int W, int col_lo, int ncols, double2* t1) {
    const int total = kh * ncols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int u = i / ncols, c = i - u * ncols;
        const long long k2 = col_lo + c;
        double re = 0, im = 0;
        for (int v = 0; v < kw; ++v) {
            if (cpos[v] < 0) continue;
            const long long m = (k2 * cpos[v]) % W;
            double sn, cs_;
            sincospi(-2.0 * (double)m / (double)W, &sn, &cs_);
            re += taps[u * kw + v] * cs_;
            im += taps[u * kw + v] * sn;
        }
        t1[i] = make_double2(re, im);
    }
}
template <typename T>
__global__ void k_taps_eK(const double2* t1, const int* rpos, int kh, int H, int ncols, cx<T>* out, SpecLayout lay, ColMap cm) {
    const size_t total = (size_t)H * ncols;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k1 = (int)(i / ncols), c = (int)(i - (size_t)k1 * ncols);
        double re = 0, im = 0;
        for (int u = 0; u < kh; ++u) {
            if (rpos[u] < 0) continue;
            const long long m = ((long long)k1 * rpos[u]) % H;
            double sn, cs_;
            sincospi(-2.0 * (double)m / (double)H, &sn, &cs_);
            const double2 a = t1[(size_t)u * ncols + c];
            re += a.x * cs_ - a.y * sn;
            im += a.x * sn + a.y * cs_;
        }
        out[spec_row(lay, 0, colpos_of_freq(cm, k1)).at(c)] = mk<T>((T)re, (T)im);
    }
}

template <typename T>
int set_kernel_local(ddn_plan* p, const double* taps, int kh, int kw) {
    const int H = p->H, W = p->W;
    const int col_lo = p->slab_lo << p->cs, ncols = std::min(p->Wh - col_lo, p->slab_n << p->cs);
    if (ncols <= 0) return DDN_OK;
    // correlation centred at (kh/2, kw/2) (eig:41-56); taps rounded to float first (eig:49), then to the plan's type
    std::vector<double> tp((size_t)kh * kw);
    for (size_t i = 0; i < tp.size(); ++i) tp[i] = (double)(T)(float)taps[i];
    std::vector<int> pos(kh + kw);
    for (int u = 0; u < kh; ++u) pos[u] = (int)((((long)kh / 2 - u) % H + H) % H);
    for (int v = 0; v < kw; ++v) pos[kh + v] = (int)((((long)kw / 2 - v) % W + W) % W);
    double* dt = nullptr;
    int* dpos = nullptr;
    double2* t1 = nullptr;
    HIPCHK(hipMalloc((void**)&dt, tp.size() * sizeof(double)));
    HIPCHK(hipMalloc((void**)&dpos, pos.size() * sizeof(int)));
    HIPCHK(hipMalloc((void**)&t1, (size_t)kh * ncols * sizeof(double2)));
    int rc = DDN_OK;
    hipError_t e = hipMemcpyAsync(dt, tp.data(), tp.size() * sizeof(double), hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dpos, pos.data(), pos.size() * sizeof(int), hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_taps_cols, dim3(ew_grid((size_t)kh * ncols)), dim3(256), 0, p->stream, dt, dpos + kh, kh, kw, W, col_lo, ncols, t1);
        SpecLayout lay = layout_of(p);
        lay.ntile = p->slab_n;
        hipLaunchKernelGGL(k_taps_eK<T>, dim3(ew_grid((size_t)H * ncols)), dim3(256), 0, p->stream, t1, dpos, kh, H, ncols, (cx<T>*)p->eK, lay, colmap<T>(p));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);      // tp / pos are read by the copies until here
    if (e != hipSuccess) rc = fail(DDN_E_HIP, hipGetErrorString(e));
    (void)hipFree(dt);
    (void)hipFree(dpos);
    (void)hipFree(t1);
    return rc;
}

template <typename T>
int set_resolvent_impl(ddn_plan* p, double cK, double cD, const void* M_dev, int cdtype) {
    const double scale = 1.0 / ((double)p->H * (double)p->W);
    if (p->restricted) {
        if (M_dev) return fail(DDN_E_UNSUPPORTED, "set_resolvent: a plan of ddn_plan_create_slab takes no injected eigen arrays");
        if (!p->have_K) return fail(DDN_E_STATE, "set_resolvent: no blur kernel set");
        const int col_lo = p->slab_lo << p->cs, ncols = std::min(p->Wh - col_lo, p->slab_n << p->cs);
        if (ncols > 0) {
            SpecLayout lay = layout_of(p);
            lay.ntile = p->slab_n;
            hipLaunchKernelGGL(k_make_G<T>, dim3(ew_grid((size_t)p->H * ncols)), dim3(256), 0, p->stream, (const cx<T>*)p->eK, (const T*)p->ed2 + col_lo, (T*)p->G,
                               p->H, ncols, lay, (T)cK, (T)cD, (T)scale);
            HIPCHK(hipGetLastError());
        }
        p->have_G = true;
        p->g_cK = cK; p->g_cD = cD;
        p->Gcm_ok = true;    // closed form of the plan's own eK: the register-resident column kernels recompute it
        return DDN_OK;
    }
    if (M_dev) {
        int rc = import_full<T>(p, M_dev, cdtype, 1, 0, scale, (cx<T>*)p->Gc, (T*)p->G);
        if (rc) return rc;
        p->have_G = true;
        p->Gcm_ok = false;
        return DDN_OK;
    }
    if (!p->have_K) return fail(DDN_E_STATE, "set_resolvent: no blur kernel set");
    const size_t total = (size_t)p->H * p->Wh;
    hipLaunchKernelGGL(k_make_G<T>, dim3(ew_grid(total)), dim3(256), 0, p->stream, (const cx<T>*)p->eK, (const T*)p->ed2, (T*)p->G,
                       p->H, p->Wh, layout_of(p), (T)cK, (T)cD, (T)scale);
    HIPCHK(hipGetLastError());
    p->have_G = true;
    p->Gcm_ok = false;
    p->g_cK = cK; p->g_cD = cD;
    p->Gcm_ok = true;  // G is the closed form of eK: the wave-per-column stage recomputes it
    return DDN_OK;
}

// Table for wcol::taps_mult (H = 4096, blur kernels of at most 31 rows): per spectral column k2 and register parity,
// 32 complex values y-coefficients c_n(k2) = sum_v tap[ph - n][v] exp(-2 pi i k2 (pw - v) / W) at slot j = n mod 32
// (n = ph - u in [-15, 15]), the odd-parity copy times w64^j (-1)^[n < 0].  Taps rounded to float first (eig:49).
int make_ctab(ddn_plan* p, const double* taps, int kh, int kw) {
    const int Wh = p->Wh, W = p->W, ph = kh / 2, pw = kw / 2;
    const long double tp = -2.0L * 3.14159265358979323846264338327950288L;
    std::vector<cx<float>> tab((size_t)Wh * 64);
    for (auto& e : tab) e = {0.f, 0.f};
    std::vector<long double> cw(kw), sw(kw);
    for (int k2 = 0; k2 < Wh; ++k2) {
        for (int v = 0; v < kw; ++v) {
            const long long m = (((long long)k2 * (pw - v)) % W + W) % W;
            cw[v] = cosl(tp * (long double)m / (long double)W);
            sw[v] = sinl(tp * (long double)m / (long double)W);
        }
        for (int u = 0; u < kh; ++u) {
            const int n = ph - u, j = ((n % 32) + 32) % 32;
            long double re = 0, im = 0;
            for (int v = 0; v < kw; ++v) {
                const long double t = (long double)(float)taps[u * kw + v];
                re += t * cw[v];
                im += t * sw[v];
            }
            tab[(size_t)k2 * 64 + j] = {(float)re, (float)im};
            const long double a = tp * (long double)j / 64.0L, sg = n < 0 ? -1.0L : 1.0L;
            const long double c = cosl(a) * sg, s_ = sinl(a) * sg;
            tab[(size_t)k2 * 64 + 32 + j] = {(float)(re * c - im * s_), (float)(re * s_ + im * c)};
        }
    }
    // column W/2: exp(-2 pi i (W/2) (pw - v) / W) = (-1)^(pw - v), so its taps along the rows are real
    std::vector<float> nt(32, 0.f);
    for (int u = 0; u < kh; ++u) {
        long double a = 0;
        for (int v = 0; v < kw; ++v) a += (long double)(float)taps[u * kw + v] * (((pw - v) & 1) ? -1.0L : 1.0L);
        nt[u] = (float)a;
    }
    if (!p->nyq_taps) HIPCHK(hipMalloc(&p->nyq_taps, 32 * sizeof(float)));
    HIPCHK(hipMemcpyAsync(p->nyq_taps, nt.data(), 32 * sizeof(float), hipMemcpyHostToDevice, p->stream));
    p->nyq_kh = kh;
    p->nyq_ph = ph;
    if (!p->ctab) HIPCHK(hipMalloc(&p->ctab, tab.size() * sizeof(cx<float>)));
    HIPCHK(hipMemcpyAsync(p->ctab, tab.data(), tab.size() * sizeof(cx<float>), hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    p->ctab_ok = true;
    return DDN_OK;
}

// While a solver loop runs entirely on the wave kernels, spectral column W/2 lives in a dense side
// array instead of the last tile (where it would use 8 bytes of every 128-byte line).
struct NyqDense {
    ddn_plan* p;
    // res2: a resolvent-mode solver (pdr, pddr, admm), whose row and column kernels both move S0 and S1
    NyqDense(ddn_plan* p_, bool wave_rows, bool res2 = false) : p(p_) {
        p->nyqd_on = false;
        p->ilv_on = false;
        if (!wave_rows) return;
        // both stages on the wave kernels: interleave the two half-spectra (not with 4-column tiles)
        p->ilv_on = res2 && (wcol_ready(p) || wcoln_ready(p)) && p->cs > 2 && p->Gcm_ok && !p->env.no_ilv;
        if (p->W != 4096 || !wcol_ready(p) || p->W % 2 || (p->Wh & 3) != 1 || p->env.no_nyq_dense) return;
        if (!p->nyqd && hipMalloc(&p->nyqd, sizeof(float) * 2 * (size_t)p->batch * p->H) != hipSuccess) { p->nyqd = nullptr; return; }
        p->nyqd_on = true;
    }
    ~NyqDense() { p->nyqd_on = false; p->ilv_on = false; }
};

// ----------------------------------------------------------------------------- solver loops
// ssim(x, y) of every image -> entry `entry` of iteration `iter`'s slot of the caller's ssim record, and (test) the
// stop test loss < tol; enqueued on the plan's stream behind the row launch that produced x, no host round trip.
template <typename T>
int ssim_step(ddn_plan* p, const void* x, const void* y, int iter, int entry, int clip_x, int first_iter, double tol, bool test) {
    if (!p->ssim_ring || iter < 1) return DDN_OK;
    if (!p->ssim_part) HIPCHK(hipMalloc((void**)&p->ssim_part, sizeof(double) * ssim_partial_count(p->H, p->W, p->batch)));
    if (!p->ssim_val) HIPCHK(hipMalloc((void**)&p->ssim_val, sizeof(double) * (size_t)p->batch));
    int rc = ssim_launch(sizeof(T) == 8 ? DDN_F64 : DDN_F32, x, y, p->H, p->W, p->batch, p->ssim_max, p->ssim_g, p->ssim_part, p->ssim_val, 1, clip_x, p->stream);
    if (rc) return rc;
    double* slot = p->ssim_ring + (size_t)(iter - 1) * p->batch * 2 + entry;
    return ssim_commit_launch(p->ssim_val, slot, 2, p->stop, p->batch, iter, first_iter, tol, test ? 1 : 0, p->stream);
}
inline int need_ssim(const ddn_plan* p, int tol_kind) {
    if (tol_kind == DDN_LOSS_SSIM && !p->ssim_ring) return fail(DDN_E_STATE, "tol_kind DDN_LOSS_SSIM needs ddn_plan_set_ssim first");
    return DDN_OK;
}

template <typename T>
int pdr_run_impl(ddn_plan* p, void* z1, void* z2, const void* b, double t, double rho, double gamma, int objective, int n_iters,
                 int tol_kind, double tol, int first_loss_iter, double* ring, int* iters_done) {
    if (!p->have_K || !p->have_G) return fail(DDN_E_STATE, "pdr_run: operators not set");
    const size_t plane = (size_t)p->batch * p->H * p->W;
    RowCfg cfg;
    const bool al = aligned16(z1) && aligned16(z2) && aligned16(b) && ((plane * sizeof(T)) % 16 == 0);
    if (!row_cfg<T>(p, al, p->batch * p->H, cfg)) return fail(DDN_E_UNSUPPORTED, "row length not supported");
    NyqDense nyq_guard(p, cfg.V == 0, true);
    typename PdrP<T>::Args a;
    RowCommon<T> rc0;
    std::memset(&a, 0, sizeof(a));
    a.z1 = (T*)z1; a.z20 = (T*)z2; a.z21 = (T*)z2 + plane; a.z22 = (T*)z2 + 2 * plane; a.b = (const T*)b;
    const bool duals_equal = (objective & DDN_OBJ_DUALS_EQUAL) != 0;
    objective &= ~DDN_OBJ_DUALS_EQUAL;
    a.t = (T)t; a.rho = (T)rho; a.tg = (T)(t * gamma); a.l2div = (T)(2 * t + 1); a.objective = objective;
    int rc;
    if ((rc = need_ssim(p, tol_kind))) return rc;
    const bool ssim = tol_kind == DDN_LOSS_SSIM;   // loss_function(z1, z1prev) = ssim: the default of primal_dr/algorithm.py:25
    const int lk = ssim ? 0 : tol_kind;
    if (ssim && !p->ssim_prev) HIPCHK(hipMalloc(&p->ssim_prev, plane * sizeof(T)));
    if ((rc = reset_stop<T>(p))) return rc;
    const cx<T>* gc = p->injected ? (const cx<T>*)p->Gc : nullptr;
    // The two difference duals (D1 == D2, mm:55-59) hold the same bits for the whole run when they start equal -- the
    // reference starts both from b (pdr:88-89) and updates them by the same expression of the same inputs.  The run then
    // points both kernel arguments at z21: every second access hits the line the first one brought in (or merges with
    // its store), so one array's worth of reads and of writes never reaches memory, and z22 is brought up to date by one
    // copy at the end.  One copy per run, so only runs of some length take it.
    // The caller states that they start equal (DDN_OBJ_DUALS_EQUAL in `objective`): nothing is compared here, and the
    // call stays free of host round trips.
    const bool twin = n_iters >= 8 && duals_equal && !p->env.no_pdr_twin;
    struct TwinBack {   // z22 <- z21 when the run ends, however it ends
        ddn_plan* p; void* dst; const void* src; size_t bytes; bool on;
        ~TwinBack() { if (on) (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, p->stream); }
    } twin_back{p, a.z22, a.z21, plane * sizeof(T), twin};
    if (twin) a.z22 = a.z21;
    if (n_iters > 0) {
        rc0 = row_common<T>(p, cfg.V, false, true);
        if ((rc = launch_solver_row<T, PdrP<T>>(p, cfg, rc0, a))) return rc;
        for (int c = 1; c <= n_iters; ++c) {
            LossFin lf = make_lf(p, ring, c - 1, lk, tol, first_loss_iter, true);
            if ((rc = col_stage<T, COL_RES2>(p, p->batch, (const cx<T>*)p->eK, (const T*)p->G, gc, 0, (T)1, lf))) return rc;
            if (ssim) HIPCHK(hipMemcpyAsync(p->ssim_prev, z1, plane * sizeof(T), hipMemcpyDeviceToDevice, p->stream));
            rc0 = row_common<T>(p, cfg.V, true, c < n_iters);
            if ((rc = launch_solver_row<T, PdrP<T>>(p, cfg, rc0, a))) return rc;
            if (ssim && (rc = ssim_step<T>(p, z1, p->ssim_prev, c, 0, 0, first_loss_iter, tol, true))) return rc;
            // ssim(z1, b): the reference logs it every 50th iteration (pdr:137-141) and records it at every iteration when it
            // saves the loss history (pdr:146-154); the cadence is the plan's (ddn_plan_set_ssim_cadence)
            if (p->ssim_ring && (c - 1) % p->ssim_b_every == 0 && (rc = ssim_step<T>(p, z1, b, c, 1, 0, 0, 0.0, false))) return rc;
        }
        if ((rc = launch_finalize<T>(p, make_lf(p, ring, n_iters, lk, tol, first_loss_iter, true)))) return rc;
    }
    return finish_iters<T>(p, iters_done, n_iters);
}

template <typename T>
int pddr_run_impl(ddn_plan* p, void* pp, void* q, const void* b, double t, double rho, double gamma, int n_iters,
                  int tol_kind, double tol, int first_loss_iter, double* ring, int* iters_done) {
    if (!p->have_K || !p->have_G) return fail(DDN_E_STATE, "pddr_run: operators not set");
    const size_t plane = (size_t)p->batch * p->H * p->W;
    RowCfg cfg;
    const bool al = aligned16(pp) && aligned16(q) && aligned16(b) && ((plane * sizeof(T)) % 16 == 0);
    if (!row_cfg<T>(p, al, p->batch * p->H, cfg)) return fail(DDN_E_UNSUPPORTED, "row length not supported");
    NyqDense nyq_guard(p, cfg.V == 0, true);
    typename PddrP<T>::Args a;
    RowCommon<T> rc0;
    std::memset(&a, 0, sizeof(a));
    a.p = (T*)pp; a.q0 = (T*)q; a.q1 = (T*)q + plane; a.q2 = (T*)q + 2 * plane; a.b = (const T*)b;
    a.t = (T)t; a.it = (T)(1.0 / t); a.rho = (T)rho; a.itg = (T)((1.0 / t) * gamma);
    int rc;
    if ((rc = need_ssim(p, tol_kind))) return rc;
    const bool ssim = tol_kind == DDN_LOSS_SSIM;
    const int lk = ssim ? 0 : tol_kind;
    if ((rc = reset_stop<T>(p))) return rc;
    const cx<T>* gc = p->injected ? (const cx<T>*)p->Gc : nullptr;
    if (n_iters > 0) {
        rc0 = row_common<T>(p, cfg.V, false, true);
        if ((rc = launch_solver_row<T, PddrP<T>>(p, cfg, rc0, a))) return rc;
        for (int c = 1; c <= n_iters; ++c) {
            LossFin lf = make_lf(p, ring, c - 1, lk, tol, first_loss_iter, true);
            if ((rc = col_stage<T, COL_RES2>(p, p->batch, (const cx<T>*)p->eK, (const T*)p->G, gc, 0, (T)1, lf))) return rc;
            rc0 = row_common<T>(p, cfg.V, true, c < n_iters);
            if ((rc = launch_solver_row<T, PddrP<T>>(p, cfg, rc0, a))) return rc;
            if (p->ssim_ring && (rc = ssim_step<T>(p, pp, b, c, 0, 1, first_loss_iter, tol, ssim))) return rc;   // loss(clip(p), b), pddr:104
        }
        if ((rc = launch_finalize<T>(p, make_lf(p, ring, n_iters, lk, tol, first_loss_iter, true)))) return rc;
    }
    return finish_iters<T>(p, iters_done, n_iters);
}

template <typename T>
int admm_run_impl(ddn_plan* p, void* u, void* w, void* y, void* z, const void* b, double t, double rho, double gamma, int objective,
                  int n_iters, int tol_kind, double tol, int first_loss_iter, double* ring, int* iters_done, void* sol) {
    if (!p->have_K || !p->have_G) return fail(DDN_E_STATE, "admm_run: operators not set");
    if (!sol) return fail(DDN_E_ARG, "admm_run: sol_out_dev is null");
    const size_t plane = (size_t)p->batch * p->H * p->W;
    RowCfg cfg;
    const bool al = aligned16(u) && aligned16(w) && aligned16(y) && aligned16(z) && aligned16(b) && aligned16(sol) && ((plane * sizeof(T)) % 16 == 0);
    if (!row_cfg<T>(p, al, p->batch * p->H, cfg, false, true)) return fail(DDN_E_UNSUPPORTED, "row length not supported");
    NyqDense nyq_guard(p, cfg.V == 0, true);
    typename AdmmP<T>::Args a;
    RowCommon<T> rc0;
    std::memset(&a, 0, sizeof(a));
    a.u = (T*)u; a.w = (T*)w;
    a.y0 = (T*)y; a.y1 = (T*)y + plane; a.y2 = (T*)y + 2 * plane;
    a.z0 = (T*)z; a.z1 = (T*)z + plane; a.z2 = (T*)z + 2 * plane;
    a.b = (const T*)b;
    a.t = (T)t; a.it = (T)(1.0 / t); a.rho = (T)rho; a.omr = (T)(1.0 - rho); a.itg = (T)((1.0 / t) * gamma);
    a.l2div = (T)(2 * (1.0 / t) + 1); a.objective = objective;
    int rc;
    if ((rc = need_ssim(p, tol_kind))) return rc;
    const bool ssim = tol_kind == DDN_LOSS_SSIM;
    const int lk = ssim ? 0 : tol_kind;
    if ((rc = reset_stop<T>(p))) return rc;
    const cx<T>* gc = p->injected ? (const cx<T>*)p->Gc : nullptr;
    // produce the transform inputs of the first solve
    rc0 = row_common<T>(p, cfg.V, false, true);
    a.sol = nullptr; a.sol_only = 0;
    if ((rc = launch_solver_row<T, AdmmP<T>>(p, cfg, rc0, a))) return rc;
    for (int c = 1; c <= n_iters; ++c) {
        // partial sums written by call c-1 describe sol of iteration c-2
        LossFin lf = make_lf(p, ring, c - 2, lk, tol, first_loss_iter, true);
        if ((rc = col_stage<T, COL_RES2>(p, p->batch, (const cx<T>*)p->eK, (const T*)p->G, gc, 0, (T)1, lf))) return rc;
        rc0 = row_common<T>(p, cfg.V, true, true);
        a.sol = (tol_kind > 0 || p->ssim_ring) ? (T*)sol : nullptr;
        if ((rc = launch_solver_row<T, AdmmP<T>>(p, cfg, rc0, a))) return rc;
        // launch c stored the look-ahead solution of iteration c - 1 (admm:118-139)
        if (p->ssim_ring && (rc = ssim_step<T>(p, sol, b, c - 1, 0, 0, first_loss_iter, tol, ssim))) return rc;
    }
    {
        LossFin lf = make_lf(p, ring, n_iters - 1, lk, tol, first_loss_iter, true);
        if ((rc = col_stage<T, COL_RES2>(p, p->batch, (const cx<T>*)p->eK, (const T*)p->G, gc, 0, (T)1, lf))) return rc;
        rc0 = row_common<T>(p, cfg.V, true, false);
        a.sol = (T*)sol; a.sol_only = 1;
        if ((rc = launch_solver_row<T, AdmmP<T>>(p, cfg, rc0, a))) return rc;
        if (n_iters >= 1) {
            if (p->ssim_ring && (rc = ssim_step<T>(p, sol, b, n_iters, 0, 0, first_loss_iter, tol, ssim))) return rc;
            if ((rc = launch_finalize<T>(p, make_lf(p, ring, n_iters, lk, tol, first_loss_iter, true)))) return rc;
        }
    }
    return finish_iters<T>(p, iters_done, n_iters);
}

template <typename T>
int cp_run_impl(ddn_plan* p, void* x, void* z, void* y, const void* b, double t, double s, double gamma, int objective, int n_iters, double* ring) {
    if (!p->have_K) return fail(DDN_E_STATE, "cp_run: operators not set");
    const size_t plane = (size_t)p->batch * p->H * p->W;
    RowCfg cfg;
    const bool al = aligned16(x) && aligned16(z) && aligned16(y) && aligned16(b) && ((plane * sizeof(T)) % 16 == 0);
    if (!row_cfg<T>(p, al, p->batch * ((p->H + 1) / 2), cfg, true)) return fail(DDN_E_UNSUPPORTED, "row length not supported");
    NyqDense nyq_guard(p, cfg.V == 0);
    // column W/2 as a short correlation along the rows inside the row kernels (RowCommon::nyq_taps): the column stage
    // then has exactly 512 quarter tiles for the chip's 512 workgroup slots instead of 513
    struct FirGuard {
        ddn_plan* p;
        ~FirGuard() { p->nyq_fir_on = false; }
    } fir_guard{p};
    p->nyq_fir_on = p->nyqd_on && p->ctab_ok && p->nyq_taps && p->nyq_kh <= 32 && sizeof(T) == 4 && !p->env.no_nyq_fir;
    int flip = 0;
    auto with_fir = [&](RowCommon<T>& rc_, int dir) {
        if (!p->nyq_fir_on) return;
        rc_.nyq_taps = (const T*)p->nyq_taps;
        rc_.nyq_kh = p->nyq_kh; rc_.nyq_ph = p->nyq_ph; rc_.nyq_dir = dir; rc_.nyq_flip = flip;
        rc_.nyq_scale = (T)(1.0 / (double)p->W);
        flip ^= 1;
    };
    CpArgs<T> a;
    RowCommon<T> rc0;
    std::memset(&a, 0, sizeof(a));
    a.x = (T*)x; a.z = (T*)z; a.y0 = (T*)y; a.y1 = (T*)y + plane; a.y2 = (T*)y + 2 * plane; a.b = (const T*)b;
    a.t = (T)t; a.s = (T)s; a.is = (T)(1.0 / s); a.isg = (T)((1.0 / s) * gamma); a.l2div = (T)(2 * (1.0 / s) + 1); a.objective = objective;
    int rc;
    if ((rc = reset_stop<T>(p))) return rc;
    const T scale = (T)(1.0 / ((double)p->H * (double)p->W));
    if (n_iters > 0) {
        a.step = 1;
        rc0 = row_common<T>(p, cfg.V, false, true);
        rc0.stop = nullptr;
        with_fir(rc0, 1);
        if ((rc = launch_cp_row<T>(p, cfg, rc0, a))) return rc;
        for (int c = 1; c <= n_iters; ++c) {
            LossFin lf = make_lf(p, ring, c - 1, 0, 0.0, 1, false);
            if ((rc = col_stage<T, COL_MUL>(p, p->batch, (const cx<T>*)p->eK, nullptr, nullptr, 0, scale, lf))) return rc;
            a.step = 0;
            rc0 = row_common<T>(p, cfg.V, true, true);
            rc0.stop = nullptr;
            with_fir(rc0, 1);     // consumes K z
            if ((rc = launch_cp_row<T>(p, cfg, rc0, a))) return rc;
            if ((rc = col_stage<T, COL_MUL>(p, p->batch, (const cx<T>*)p->eK, nullptr, nullptr, 1, scale, no_loss()))) return rc;
            a.step = 1;
            rc0 = row_common<T>(p, cfg.V, true, c < n_iters);
            rc0.stop = nullptr;
            with_fir(rc0, -1);    // consumes K^T y0
            if ((rc = launch_cp_row<T>(p, cfg, rc0, a))) return rc;
            if (p->ssim_ring && (rc = ssim_step<T>(p, x, b, c, 0, 0, 0, 0.0, false))) return rc;   // loss(x_next, b), cp:96
        }
        if ((rc = launch_finalize<T>(p, make_lf(p, ring, n_iters, 0, 0.0, 1, false)))) return rc;
    }
    return DDN_OK;
}

template <typename T>
int apply_impl(ddn_plan* p, int op, const void* x, void* y) {
    const size_t n = (size_t)p->batch * p->H * p->W;
    const size_t total = (size_t)p->H * p->Wh;
    switch (op) {
        case DDN_OP_K:
        case DDN_OP_KT:
            if (!p->have_K) return fail(DDN_E_STATE, "apply: no blur kernel set");
            return apply_mult<T>(p, (const cx<T>*)p->eK, op == DDN_OP_KT, (const T*)x, (T*)y, p->batch);
        case DDN_OP_D:
        case DDN_OP_DT:
            if (p->injected) {
                hipLaunchKernelGGL(k_bcast_row<T>, dim3(ew_grid(total)), dim3(256), 0, p->stream, (const cx<T>*)p->eDrow, (cx<T>*)p->tmp_mult, p->H, p->Wh, layout_of(p));
                HIPCHK(hipGetLastError());
                return apply_mult<T>(p, (const cx<T>*)p->tmp_mult, op == DDN_OP_DT, (const T*)x, (T*)y, p->batch);
            }
            if (x == y) return fail(DDN_E_ARG, "apply D/DT: in-place not supported");
            hipLaunchKernelGGL(k_stencil<T>, dim3(ew_grid(n)), dim3(256), 0, p->stream, (const T*)x, (T*)y, p->W, n, op == DDN_OP_DT ? 1 : 0);
            HIPCHK(hipGetLastError());
            return DDN_OK;
        case DDN_OP_RESOLVENT: {
            if (!p->have_G) return fail(DDN_E_STATE, "apply: resolvent not set");
            int rc;
            if ((rc = rows_r2c<T>(p, (const T*)x, p->batch))) return rc;
            if ((rc = col_stage<T, COL_MULR>(p, p->batch, nullptr, (const T*)p->G, p->injected ? (const cx<T>*)p->Gc : nullptr, 0, (T)1, no_loss()))) return rc;
            return rows_c2r<T>(p, (T*)y, p->batch, (T)1, nullptr);
        }
        case DDN_OP_MAT: {
            if (!p->have_K) return fail(DDN_E_STATE, "apply: no blur kernel set");
            hipLaunchKernelGGL(k_make_mat<T>, dim3(ew_grid(total)), dim3(256), 0, p->stream, (const cx<T>*)p->eK, (const T*)p->ed2, (cx<T>*)p->tmp_mult, p->H, p->Wh, layout_of(p));
            HIPCHK(hipGetLastError());
            return apply_mult<T>(p, (const cx<T>*)p->tmp_mult, 0, (const T*)x, (T*)y, p->batch);
        }
    }
    return fail(DDN_E_ARG, "apply: unknown operator");
}

// ------------------------------------------------ one image split into row slabs over several GPUs
// The plan is created with the full image size (its multipliers cover every frequency); a rank owns
// `rows` consecutive image rows and a range of spectral tiles.  The fused row kernels run on the
// local rows with a caller-owned half-spectrum buffer laid out [tile][2 spectra][rows][C], so that
// the tiles of one destination rank are one contiguous block of it; the column stage runs on the
// local tile range of a caller-owned [2 spectra][local tiles][H][C] buffer.  The exchange between
// the two buffers (an all-to-all) belongs to the caller.
struct SlabRows {
    ddn_plan* p;
    int H;
    SlabRows(ddn_plan* q, int rows) : p(q), H(q->H) { p->H = rows; }
    ~SlabRows() { p->H = H; }
};

template <typename T>
int slab_rows_impl(ddn_plan* p, int solver, int rows, void* spec, void* s0, void* s1, void* s2, void* s3, const void* b,
                   double t, double rho, double gamma, int objective, int consume, int produce, void* sol, int sol_only) {
    if (!p->have_K || !p->have_G) return fail(DDN_E_STATE, "slab_rows: operators not set");
    SlabRows guard(p, rows);  // row_cfg / row_common size the launch from p->H
    const size_t plane = (size_t)rows * p->W;
    RowCfg cfg;
    const bool al = aligned16(s0) && aligned16(s1) && (!s2 || aligned16(s2)) && (!s3 || aligned16(s3)) && aligned16(b) && (!sol || aligned16(sol)) &&
                    ((plane * sizeof(T)) % 16 == 0);
    if (!row_cfg<T>(p, al, rows, cfg, false, solver == 0)) return fail(DDN_E_UNSUPPORTED, "row length not supported");
    RowCommon<T> c = row_common<T>(p, cfg.V, consume != 0, produce != 0);   // c.stop: the plan's flag, see ddn_slab_begin / ddn_slab_stop_test
    const size_t C = (size_t)1 << p->cs;
    c.S0 = (cx<T>*)spec;
    c.S1 = (cx<T>*)spec + (size_t)rows * C;
    c.lay.H = 2 * rows;  // tile pitch in rows: S0 rows then S1 rows of a tile
    c.nyqd = nullptr;
    c.ilv = 0;
    if (solver == 0) {
        typename AdmmP<T>::Args a;
        std::memset(&a, 0, sizeof(a));
        a.u = (T*)s0; a.w = (T*)s1;
        a.y0 = (T*)s2; a.y1 = (T*)s2 + plane; a.y2 = (T*)s2 + 2 * plane;
        a.z0 = (T*)s3; a.z1 = (T*)s3 + plane; a.z2 = (T*)s3 + 2 * plane;
        a.b = (const T*)b;
        a.t = (T)t; a.it = (T)(1.0 / t); a.rho = (T)rho; a.omr = (T)(1.0 - rho); a.itg = (T)((1.0 / t) * gamma);
        a.l2div = (T)(2 * (1.0 / t) + 1); a.objective = objective;
        a.sol = (T*)sol; a.sol_only = sol_only;
        if (!s2 || !s3) return fail(DDN_E_ARG, "slab_rows: admm needs u, w, y, z");
        return launch_solver_row<T, AdmmP<T>>(p, cfg, c, a);
    }
    if (solver == 1) {
        typename PdrP<T>::Args a;
        std::memset(&a, 0, sizeof(a));
        a.z1 = (T*)s0; a.z20 = (T*)s1; a.z21 = (T*)s1 + plane; a.z22 = (T*)s1 + 2 * plane; a.b = (const T*)b;
        a.t = (T)t; a.rho = (T)rho; a.tg = (T)(t * gamma); a.l2div = (T)(2 * t + 1); a.objective = objective;
        return launch_solver_row<T, PdrP<T>>(p, cfg, c, a);
    }
    if (solver == 2) {
        typename PddrP<T>::Args a;
        std::memset(&a, 0, sizeof(a));
        a.p = (T*)s0; a.q0 = (T*)s1; a.q1 = (T*)s1 + plane; a.q2 = (T*)s1 + 2 * plane; a.b = (const T*)b;
        a.t = (T)t; a.it = (T)(1.0 / t); a.rho = (T)rho; a.itg = (T)((1.0 / t) * gamma);
        return launch_solver_row<T, PddrP<T>>(p, cfg, c, a);
    }
    return fail(DDN_E_ARG, "slab_rows: unknown solver");
}

// Column stage on tiles [tile_lo, tile_lo + ntiles) of the image's half-spectra: MODE COL_RES2 (the
// resolvent solvers; spec holds both spectra) or COL_MUL (Chambolle-Pock: one spectrum times eK or conj(eK)).
template <typename T, int MODE>
int slab_cols_impl(ddn_plan* p, void* spec_all, int tile_lo_all, int ntiles_all, int conj_mult, int nblocks, int first, int count) {
    if (!p->have_K || (MODE == COL_RES2 && !p->have_G)) return fail(DDN_E_STATE, "slab_cols: operators not set");
    // the buffer holds ntiles_all tiles from tile_lo_all on; this call works on `count` of them from the first-th on
    if (first < 0 || count < 0 || first + count > ntiles_all) return fail(DDN_E_ARG, "slab_cols: part outside the buffer's tiles");
    if (count == 0) return DDN_OK;
    const int tile_lo = tile_lo_all + first, ntiles = count;
    if (p->restricted && (tile_lo < p->slab_lo || tile_lo + ntiles > p->slab_lo + p->slab_n))
        return fail(DDN_E_ARG, "slab_cols: tiles outside the range given to ddn_slab_restrict");
    constexpr int NS = MODE == COL_RES2 ? 2 : 1;
    const T scale = MODE == COL_MUL ? (T)(1.0 / ((double)p->H * (double)p->W)) : (T)1;
    const size_t C = (size_t)1 << p->cs;
    // nblocks = 0: spec is [spectrum][tile][H][C].  nblocks = P >= 1: spec is [source rank][tile][spectrum][H / P][C],
    // what an all-to-all of the ranks' row-side buffers delivers; the kernels then address a tile with pitch
    // NS * rows, find the second spectrum rows * C elements after the first, and displace the rows of source
    // rank i by i * (ntiles * NS - 1) * rows rows.
    int pitch = 0, sh = 0;
    long long extra_rows = 0;
    size_t s1_off = (size_t)ntiles_all * ((size_t)p->H << p->cs);
    if (nblocks >= 1) {
        if (p->H % nblocks) return fail(DDN_E_ARG, "slab_cols: H is not a multiple of the number of row blocks");
        const int rows = p->H / nblocks;
        pitch = NS * rows;
        s1_off = (size_t)rows * C;
        extra_rows = nblocks > 1 ? (long long)rows * (ntiles_all * NS - 1) : 0;
        if (extra_rows) {
            if (rows & (rows - 1)) return fail(DDN_E_UNSUPPORTED, "slab_cols: rows per block must be a power of two");
            while ((1 << sh) < rows) ++sh;
        }
    }
    void* const spec = (cx<T>*)spec_all + (size_t)first * ((size_t)(pitch ? pitch : p->H) << p->cs);   // tile `first` of the buffer
    if constexpr (sizeof(T) == 4) {
        // H = 4096 / 512-2048: the register-resident column kernels, pointed at the local columns (their
        // multipliers are column-major, so a column range is a pointer offset; column W/2 stays in its tile)
        const bool wave1 = wcol_ready(p), waven = !wave1 && wcoln_ready(p);
        if ((wave1 || waven) && (MODE == COL_MUL || p->Gcm_ok) && p->cs > 2 && (!extra_rows || sh >= 6)) {
            const int col_lo = tile_lo << p->cs;
            wcol::Args w;
            std::memset(&w, 0, sizeof(w));
            w.S0 = (cx<float>*)spec;
            w.S1 = (cx<float>*)spec + s1_off;
            // a plan of ddn_plan_create_slab holds the column-major eigenvalues of its own columns only
            w.eKc = (const cx<float>*)p->eKc + (size_t)(col_lo - (p->restricted ? (p->slab_lo << p->cs) : 0)) * wcol::kColPitch;
            w.ctab = (wave1 && p->ctab_ok) ? (const cx<float>*)p->ctab + (size_t)col_lo * 64 : nullptr;
            w.wtw = (const cx<float>*)(wave1 ? p->wtw : p->wtwH);
            w.ed2 = (const float*)p->ed2 + col_lo;
            w.cK = (float)p->g_cK; w.cD = (float)p->g_cD; w.gscale = (float)(1.0 / ((double)p->H * (double)p->W));
            w.lay = layout_of(p);
            w.lay.ntile = ntiles;
            if (pitch) w.lay.H = pitch;
            w.Hfft = p->H;
            w.rb_sh6 = extra_rows ? sh - 6 : 0;
            w.rb_extra = extra_rows * (long long)C;
            w.Wh = std::min(p->Wh - col_lo, ntiles << p->cs);
            w.batch = 1;
            w.conj_mult = conj_mult;
            w.scale = (float)scale;
            w.lf = no_loss();
            TimerScope ts(p, false);
            return waven ? wcoln_launch(p->stream, MODE, w) : wcol_launch(p->stream, MODE, w);
        }
    }
    // the plan's own fields are pointed at the local tile range for the duration of the call; the
    // multipliers are offset to the same tiles, and the tile-generic column kernels are selected
    struct Swap {
        ddn_plan* p; void* S0; void* S1; int ntile;
        explicit Swap(ddn_plan* q) : p(q), S0(q->S0), S1(q->S1), ntile(q->ntile) { p->slab_cols = true; }
        ~Swap() { p->S0 = S0; p->S1 = S1; p->ntile = ntile; p->slab_cols = false; p->slab_pitch = 0; p->rb_shift = 0; p->rb_extra = 0; }
    } guard(p);
    const size_t tile_elems = ((size_t)(p->H + p->tile_pad_rows)) << p->cs;
    p->S0 = spec;
    p->S1 = (cx<T>*)spec + s1_off;
    p->ntile = ntiles;
    p->slab_pitch = pitch;
    p->rb_shift = sh;
    p->rb_extra = extra_rows;
    const size_t off = (size_t)(tile_lo - p->slab_lo) * tile_elems;   // slab_lo: first tile the plan holds multipliers of (0 unless ddn_slab_restrict)
    if (MODE == COL_MUL) return col_stage<T, COL_MUL>(p, 1, (const cx<T>*)p->eK + off, nullptr, nullptr, conj_mult, scale, no_loss());
    return col_stage<T, COL_RES2>(p, 1, (const cx<T>*)p->eK + off, (const T*)p->G + off, p->injected ? (const cx<T>*)p->Gc + off : nullptr, 0, (T)1, no_loss());
}

// Chambolle-Pock row step (cp:58-92) on the local rows; one half-spectrum, laid out [tile][rows][C].
template <typename T>
int slab_rows_cp_impl(ddn_plan* p, int rows, void* spec, void* spec_aux, void* x, void* z, void* y, const void* b, double t, double s, double gamma,
                      int objective, int step, int consume, int produce) {
    if (!p->have_K) return fail(DDN_E_STATE, "slab_rows_cp: operators not set");
    // injected eigen arrays: D^T y travels from the dual-step launch to the primal-step launch of the same
    // rows as a spectrum of its own, which never passes through the column stage
    if (p->injected && !spec_aux) return fail(DDN_E_ARG, "slab_rows_cp: injected eigen arrays need spec_aux_dev");
    SlabRows guard(p, rows);
    const size_t plane = (size_t)rows * p->W;
    RowCfg cfg;
    const bool al = aligned16(x) && aligned16(z) && aligned16(y) && aligned16(b) && ((plane * sizeof(T)) % 16 == 0);
    if (!row_cfg<T>(p, al, (rows + 1) / 2, cfg, true)) return fail(DDN_E_UNSUPPORTED, "row length not supported");
    CpArgs<T> a;
    std::memset(&a, 0, sizeof(a));
    a.x = (T*)x; a.z = (T*)z; a.y0 = (T*)y; a.y1 = (T*)y + plane; a.y2 = (T*)y + 2 * plane; a.b = (const T*)b;
    a.t = (T)t; a.s = (T)s; a.is = (T)(1.0 / s); a.isg = (T)((1.0 / s) * gamma); a.l2div = (T)(2 * (1.0 / s) + 1); a.objective = objective;
    a.step = step;
    RowCommon<T> c = row_common<T>(p, cfg.V, consume != 0, produce != 0);
    c.stop = nullptr;
    c.S0 = (cx<T>*)spec;
    c.S1 = (cx<T>*)spec_aux;
    c.lay.H = rows;
    c.nyqd = nullptr;
    c.ilv = 0;
    return launch_cp_row<T>(p, cfg, c, a);
}

// raw loss sums of a slab's rows: sum |d| and sum d^2 of the primary pair, then of the secondary pair
__global__ void k_slab_sums(const double* partials, int rows, double* sums4) {
    __shared__ double red[16 * 4];
    double v[4] = {0, 0, 0, 0};
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] += partials[(size_t)r * 4 + s];
    }
    block_reduce<4>(v, red, threadIdx.x, blockDim.x);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) sums4[s] = v[s];
    }
}
// whole-image record and stop test from the sums over all slabs (the same arithmetic as loss_finalize)
__global__ void k_slab_stop_test(const double* sums4, int kind, double tol, int iter, int first_iter, double npix, int* stop, double* record8) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (stop[0] >= 0) return;  // frozen: keeps its last record
    double out[8];
    for (int h = 0; h < 2; ++h) {
        const double s1 = sums4[2 * h], s2 = sums4[2 * h + 1];
        const double mse = s2 / npix;
        out[4 * h + 0] = s1;
        out[4 * h + 1] = sqrt(s2);
        out[4 * h + 2] = mse;
        out[4 * h + 3] = (mse == 0.0) ? (double)INFINITY : 10.0 * log10(1.0 / mse);
    }
    if (record8) {
        for (int i = 0; i < 8; ++i) record8[i] = out[i];
    }
    if (kind >= 1 && kind <= 4 && iter >= first_iter && out[kind - 1] < tol) stop[0] = iter;
}

}  // namespace

// ------------------------------------------------- scratch of the allocation-free entry point
// ddn_reduce_static keeps its per-chunk partial sums here instead of in a hipMalloc'ed buffer:
// [image][chunk][2] doubles, 512 KiB, resident on every GPU the library runs on until the process
// ends.  In host code the array's name is only a handle: the device address comes from
// hipGetSymbolAddress, once per device, and that pointer is what kernels and copies are given.
namespace {
constexpr int kStaticChunks = 256;
constexpr int kMaxDevices = 64;
}  // namespace
static __device__ __align__(256) double scratch[2 * kStaticChunks * DDN_REDUCE_STATIC_MAX_BATCH];
namespace {
double* g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mu;

// device address of `scratch` on the current device (resolved at the first call on that device)
int static_scratch(double** out) {
    int dev = -1;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(DDN_E_UNSUPPORTED, "static scratch: device ordinal out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) return fail(DDN_E_HIP, e != hipSuccess ? hipGetErrorString(e) : "hipGetSymbolAddress returned null");
        g_scratch_ptr[dev] = (double*)ptr;
    }
    *out = g_scratch_ptr[dev];
    return DDN_OK;
}
}  // namespace

// =============================================================================== C ABI
extern "C" {

int ddn_version(void) { return 320; }
const char* ddn_last_error(void) { return ddn_err().c_str(); }

static int plan_create_impl(ddn_plan** out, int H, int W, int batch, int dtype, void* stream, bool restricted) {
    if (!out || H < 1 || W < 2 || batch < 1 || (dtype != DDN_F32 && dtype != DDN_F64)) return fail(DDN_E_ARG, "plan_create: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DDN_E_HIP, "plan_create: no HIP device");
    ddn_plan* p = new ddn_plan();
    p->H = H; p->W = W; p->batch = batch; p->dtype = dtype; p->stream = (hipStream_t)stream;
    p->Wh = W / 2 + 1;
    p->restricted = restricted;
    p->env.read();
    (void)hipGetDevice(&p->device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess) p->num_cu = prop.multiProcessorCount;
    if (!factorize(W, p->fw) || !factorize(H, p->fh)) {
        delete p;
        return fail(DDN_E_UNSUPPORTED, "plan_create: image size has a prime factor > 8191");
    }
    const bool ok = dtype == DDN_F32 ? choose_col<float>(p) : choose_col<double>(p);
    if (!ok) {
        delete p;
        return fail(DDN_E_UNSUPPORTED, "plan_create: column transform does not fit LDS");
    }
    if (p->A == 1) p->fb = p->fh;
    const size_t rl = dtype == DDN_F32 ? row_lds_bytes<float>(W, 1) : row_lds_bytes<double>(W, 1);
    if (rl > kLdsMax || getenv("DDN_FORCE_LONG_ROWS")) {
        // the row transform is split W = R x M (ddn_long.hpp); DDN_FORCE_LONG_ROWS: the same path on short rows, for tests
        const int forced = getenv("DDN_FORCE_LONG_ROWS") ? atoi(getenv("DDN_FORCE_LONG_ROWS")) : 0;   // 2, 4, 8, 16: that split; else the smallest
        for (int R : {2, 4, 8, 16}) {
            if (forced >= 2 && R != forced) continue;
            if (W % R == 0 && W / R <= 4096 && W / R >= 2 && factorize(W / R, p->fl) && fft_min_threads(p->fl) <= 256) {   // (the split-row kernels run 256 threads)
                p->long_R = R;
                p->long_M = W / R;
                break;
            }
        }
        if (!p->long_R) {
            delete p;
            return fail(DDN_E_UNSUPPORTED, "plan_create: row length too large (W must be 2, 4, 8 or 16 times a length of at most 4096 with prime factors <= 509: the split-row kernels run 256 threads)");
        }
    }
    int rc = dtype == DDN_F32 ? plan_alloc<float>(p) : plan_alloc<double>(p);
    if (rc) {
        ddn_plan_destroy(p);
        return rc;
    }
    {
        // set-up for ddn_reduce_static on this device, so that its first call has nothing left to resolve
        double* unused = nullptr;
        (void)static_scratch(&unused);
    }
    *out = p;
    return DDN_OK;
}

int ddn_plan_create(ddn_plan** out, int H, int W, int batch, int dtype, void* stream) { return plan_create_impl(out, H, W, batch, dtype, stream, false); }

int ddn_plan_create_slab(ddn_plan** out, int H, int W, int dtype, void* stream) { return plan_create_impl(out, H, W, 1, dtype, stream, true); }

int ddn_slab_restrict(ddn_plan* p, int tile_lo, int n_tiles_local) {
    if (!p) return fail(DDN_E_ARG, "null plan");
    if (!p->restricted) return fail(DDN_E_STATE, "slab_restrict: not a plan of ddn_plan_create_slab");
    if (tile_lo < 0 || n_tiles_local < 0 || tile_lo + n_tiles_local > p->ntile) return fail(DDN_E_ARG, "slab_restrict: tile range outside the plan");
    for (void** q : {&p->eK, &p->G}) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
    }
    p->slab_lo = tile_lo;
    p->slab_n = n_tiles_local;
    p->have_K = p->have_G = false;
    p->eKc_ok = p->Gcm_ok = p->ctab_ok = false;   // the column-major copy belongs to the previous range
    const size_t m = std::max<size_t>(1, (size_t)n_tiles_local * (((size_t)(p->H + p->tile_pad_rows)) << p->cs));
    const size_t esz = p->dtype == DDN_F32 ? sizeof(float) : sizeof(double);
    HIPCHK(hipMalloc(&p->eK, m * 2 * esz));
    HIPCHK(hipMalloc(&p->G, m * esz));
    HIPCHK(hipMemsetAsync(p->eK, 0, m * 2 * esz, p->stream));
    HIPCHK(hipMemsetAsync(p->G, 0, m * esz, p->stream));
    return DDN_OK;
}

// the entries that work on whole planes held by the plan refuse a plan that holds none
#define DDN_WHOLE_PLANE(p, name) \
    if ((p) && (p)->restricted) return fail(DDN_E_STATE, name ": this plan holds only the multipliers of a tile range (ddn_plan_create_slab); use the ddn_slab_* entries")

int ddn_plan_destroy(ddn_plan* p) {
    if (!p) return DDN_OK;
    void* bufs[] = {p->twW, p->wtw, p->wtwW, p->wtwH, p->twL, p->twHalf, p->edL, p->Ylong, p->nyqd, p->ssim_part, p->ssim_val, p->ssim_prev, p->nyq_taps, p->eKc, p->ctab, p->Gcm, p->twH, p->twA, p->twB, p->S0, p->eK, p->G, p->Gc, p->eDrow, p->ed2, p->tmp_mult, p->partials, p->stop};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    for (auto& e : p->ev_pool) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    delete p;
    return DDN_OK;
}

int ddn_plan_set_stream(ddn_plan* p, void* stream) {
    if (!p) return fail(DDN_E_ARG, "null plan");
    p->stream = (hipStream_t)stream;
    return DDN_OK;
}

int ddn_plan_tune(ddn_plan* p, int cols_per_block, int split_A, int col_threads) {
    DDN_WHOLE_PLANE(p, "plan_tune");
    if (!p) return fail(DDN_E_ARG, "null plan");
    if (p->have_K) return fail(DDN_E_STATE, "plan_tune must precede set_kernel / set_eig");
    if (split_A > 0 && p->H % split_A) return fail(DDN_E_ARG, "plan_tune: split_A does not divide H");
    p->tune_cols = cols_per_block;
    p->tune_A = split_A;
    p->col_threads = col_threads;
    const bool ok = p->dtype == DDN_F32 ? choose_col<float>(p) : choose_col<double>(p);
    if (!ok) return fail(DDN_E_UNSUPPORTED, "plan_tune: tiling does not fit LDS");
    if (p->A == 1) p->fb = p->fh;
    int rc;
    if (p->dtype == DDN_F32) {
        if ((rc = alloc_col_tables<float>(p))) return rc;
        return alloc_spec<float>(p);
    }
    if ((rc = alloc_col_tables<double>(p))) return rc;
    return alloc_spec<double>(p);
}

int ddn_plan_set_kernel(ddn_plan* p, const double* taps, int kh, int kw) {
    if (!p || !taps || kh < 1 || kw < 1) return fail(DDN_E_ARG, "set_kernel: bad arguments");
    if (p->restricted) {
        if (!p->eK) return fail(DDN_E_STATE, "set_kernel: call ddn_slab_restrict first");
        int rcl = p->dtype == DDN_F32 ? set_kernel_local<float>(p, taps, kh, kw) : set_kernel_local<double>(p, taps, kh, kw);
        if (rcl) return rcl;
        p->have_K = true;
        p->injected = false;
        p->eKc_ok = p->Gcm_ok = p->ctab_ok = false;
        // the register-resident column kernels (H = 512 ... 4096, float) read a column-major copy of the eigenvalues:
        // here of the plan's own columns only (ddn_slab_cols offsets into it relative to the first of them)
        if (p->dtype == DDN_F32 && !p->env.no_wave_cols &&
            ((p->wtw && p->H == 4096 && p->cs == 4) || (p->wtwH && (1 << p->cs) >= 4096 / p->H && p->cs <= 4))) {
            const int col_lo = p->slab_lo << p->cs, ncols = std::min(p->Wh - col_lo, p->slab_n << p->cs);
            if (ncols > 0) {
                if (p->eKc) {
                    HIPCHK(hipStreamSynchronize(p->stream));
                    (void)hipFree(p->eKc);
                    p->eKc = nullptr;
                }
                HIPCHK(hipMalloc(&p->eKc, (size_t)wcol::kColPitch * ncols * sizeof(cx<float>)));
                SpecLayout lay = layout_of(p);
                lay.ntile = p->slab_n;
                hipLaunchKernelGGL(wcol::k_mult_colmajor<float>, dim3(ew_grid((size_t)p->H * ncols)), dim3(256), 0, p->stream, (const cx<float>*)p->eK,
                                   (const float*)nullptr, (cx<float>*)p->eKc, (float*)nullptr, p->H, ncols, lay, colmap<float>(p));
                HIPCHK(hipGetLastError());
                p->eKc_ok = true;
                if (p->wtw && p->H == 4096 && p->cs == 4 && kh <= 31 && !p->env.no_taps) {
                    int rct = make_ctab(p, taps, kh, kw);     // every column of the plane: 512 bytes each
                    if (rct) return rct;
                }
            }
        }
        return DDN_OK;
    }
    int rc = p->dtype == DDN_F32 ? set_kernel_impl<float>(p, taps, kh, kw, false, (cx<float>*)p->eK)
                                 : set_kernel_impl<double>(p, taps, kh, kw, false, (cx<double>*)p->eK);
    if (rc) return rc;
    p->have_K = true;
    p->injected = false;
    p->eKc_ok = p->Gcm_ok = false;
    p->ctab_ok = false;
    if (p->dtype == DDN_F32 && p->wtw && p->H == 4096 && p->cs == 4 && kh <= 31 && !p->env.no_taps) {
        if ((rc = make_ctab(p, taps, kh, kw))) return rc;
    }
    if (p->dtype == DDN_F32 && ((p->wtw && p->H == 4096 && p->cs == 4) || (p->wtwH && (1 << p->cs) >= 4096 / p->H && p->cs <= 4))) {
        if ((rc = make_colmajor<float>(p, (const cx<float>*)p->eK, nullptr, &p->eKc, nullptr))) return rc;
        p->eKc_ok = true;
    }
    return DDN_OK;
}

int ddn_plan_set_eig(ddn_plan* p, const void* eigK, const void* eigD, int cdtype) {
    DDN_WHOLE_PLANE(p, "set_eig");
    if (!p || !eigK) return fail(DDN_E_ARG, "set_eig: bad arguments");
    int rc;
    if (p->dtype == DDN_F32) rc = import_full<float>(p, eigK, cdtype, 0, 0, 1.0, (cx<float>*)p->eK, nullptr);
    else rc = import_full<double>(p, eigK, cdtype, 0, 0, 1.0, (cx<double>*)p->eK, nullptr);
    if (rc) return rc;
    p->have_K = true;
    p->eKc_ok = p->Gcm_ok = p->ctab_ok = false;
    // row-spectral D table: Hermitian-symmetrised first row of eigD, or the analytic 1 - w^k
    const int W = p->W, Wh = p->Wh;
    std::vector<double> row(2 * (size_t)W);
    if (eigD) {
        if (cdtype == DDN_C64) {
            std::vector<float> t(2 * (size_t)W);
            HIPCHK(hipMemcpy(t.data(), eigD, sizeof(float) * 2 * W, hipMemcpyDeviceToHost));
            for (int i = 0; i < 2 * W; ++i) row[i] = t[i];
        } else {
            HIPCHK(hipMemcpy(row.data(), eigD, sizeof(double) * 2 * W, hipMemcpyDeviceToHost));
        }
    } else {
        for (int k = 0; k < W; ++k) {
            const long double ang = -2.0L * 3.14159265358979323846264338327950288L * k / W;
            row[2 * k] = (double)(1.0L - cosl(ang));
            row[2 * k + 1] = (double)(-sinl(ang));
        }
    }
    std::vector<double> sym(2 * (size_t)p->Wh, 0.0);
    for (int k = 0; k < Wh; ++k) {
        const int km = k == 0 ? 0 : W - k;
        sym[2 * k] = 0.5 * (row[2 * k] + row[2 * km]);
        sym[2 * k + 1] = 0.5 * (row[2 * k + 1] - row[2 * km + 1]);
    }
    if (p->dtype == DDN_F32) {
        std::vector<float> f(sym.begin(), sym.end());
        HIPCHK(hipMemcpy(p->eDrow, f.data(), sizeof(float) * 2 * p->Wh, hipMemcpyHostToDevice));
    } else {
        HIPCHK(hipMemcpy(p->eDrow, sym.data(), sizeof(double) * 2 * p->Wh, hipMemcpyHostToDevice));
    }
    p->injected = true;
    return DDN_OK;
}

int ddn_plan_set_resolvent(ddn_plan* p, double cK, double cD, const void* M_dev, int cdtype) {
    if (!p) return fail(DDN_E_ARG, "null plan");
    if (p->injected && !M_dev) return fail(DDN_E_ARG, "set_resolvent: injected eigen arrays need the injected eig_vals_mat too");
    return p->dtype == DDN_F32 ? set_resolvent_impl<float>(p, cK, cD, M_dev, cdtype) : set_resolvent_impl<double>(p, cK, cD, M_dev, cdtype);
}

int ddn_plan_set_ssim(ddn_plan* p, const float* g11_host, double max_val, double* ssim_ring_dev) {
    if (!p) return fail(DDN_E_ARG, "null plan");
    if (!ssim_ring_dev) {
        p->ssim_ring = nullptr;
        return DDN_OK;
    }
    if (!g11_host) return fail(DDN_E_ARG, "set_ssim: window taps missing");
    for (int i = 0; i < 11; ++i) p->ssim_g[i] = g11_host[i];
    p->ssim_max = max_val;
    p->ssim_ring = ssim_ring_dev;
    return DDN_OK;
}

int ddn_plan_set_ssim_cadence(ddn_plan* p, int every) {
    if (!p || every < 1) return fail(DDN_E_ARG, "set_ssim_cadence: bad arguments");
    p->ssim_b_every = every;
    return DDN_OK;
}

int ddn_plan_get_eigK(ddn_plan* p, void* out_dev) {
    DDN_WHOLE_PLANE(p, "get_eigK");
    if (!p || !out_dev) return fail(DDN_E_ARG, "get_eigK: bad arguments");
    if (!p->have_K) return fail(DDN_E_STATE, "get_eigK: no blur kernel set");
    const size_t total = (size_t)p->H * p->W;
    if (p->dtype == DDN_F32)
        hipLaunchKernelGGL(k_export_full<float>, dim3(ew_grid(total)), dim3(256), 0, p->stream, (const cx<float>*)p->eK, (cx<float>*)out_dev, p->H, p->W, layout_of(p), colmap<float>(p));
    else
        hipLaunchKernelGGL(k_export_full<double>, dim3(ew_grid(total)), dim3(256), 0, p->stream, (const cx<double>*)p->eK, (cx<double>*)out_dev, p->H, p->W, layout_of(p), colmap<double>(p));
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

int ddn_apply(ddn_plan* p, int op, const void* x, void* y) {
    DDN_WHOLE_PLANE(p, "apply");
    if (!p || !x || !y) return fail(DDN_E_ARG, "apply: bad arguments");
    return p->dtype == DDN_F32 ? apply_impl<float>(p, op, x, y) : apply_impl<double>(p, op, x, y);
}

int ddn_apply_multiplier(ddn_plan* p, const void* mult, int cdtype, int flags, const void* x, void* y) {
    DDN_WHOLE_PLANE(p, "apply_multiplier");
    if (!p || !mult || !x || !y) return fail(DDN_E_ARG, "apply_multiplier: bad arguments");
    int rc;
    if (p->dtype == DDN_F32) {
        if ((rc = import_full<float>(p, mult, cdtype, (flags >> 1) & 1, flags & 1, 1.0, (cx<float>*)p->tmp_mult, nullptr))) return rc;
        return apply_mult<float>(p, (const cx<float>*)p->tmp_mult, 0, (const float*)x, (float*)y, p->batch);
    }
    if ((rc = import_full<double>(p, mult, cdtype, (flags >> 1) & 1, flags & 1, 1.0, (cx<double>*)p->tmp_mult, nullptr))) return rc;
    return apply_mult<double>(p, (const cx<double>*)p->tmp_mult, 0, (const double*)x, (double*)y, p->batch);
}

int ddn_circular_convolve2d(ddn_plan* p, const double* taps, int kh, int kw, const void* image, void* out) {
    DDN_WHOLE_PLANE(p, "circular_convolve2d");
    if (!p || !taps || !image || !out) return fail(DDN_E_ARG, "circular_convolve2d: bad arguments");
    if (kh > p->H || kw > p->W) return fail(DDN_E_ARG, "circular_convolve2d: kernel larger than the image");
    int rc;
    if (p->dtype == DDN_F32) {
        if ((rc = set_kernel_impl<float>(p, taps, kh, kw, true, (cx<float>*)p->tmp_mult))) return rc;
        return apply_mult<float>(p, (const cx<float>*)p->tmp_mult, 0, (const float*)image, (float*)out, p->batch);
    }
    if ((rc = set_kernel_impl<double>(p, taps, kh, kw, true, (cx<double>*)p->tmp_mult))) return rc;
    return apply_mult<double>(p, (const cx<double>*)p->tmp_mult, 0, (const double*)image, (double*)out, p->batch);
}

int ddn_prox(int kind, int dtype, const void* x, void* y, size_t n, double lambda, void* stream) {
    if (!x || !y || kind < 0 || kind > 4) return fail(DDN_E_ARG, "prox: bad arguments");
    if (n == 0) return DDN_OK;
    hipStream_t st = (hipStream_t)stream;
    double* norm2 = nullptr;
    if (kind == DDN_PROX_L2) {
        // whole-tensor norm first (prox:14)
        double* part = nullptr;
        const int chunks = 256;
        HIPCHK(hipMalloc((void**)&part, sizeof(double) * (2 * chunks + 4)));
        if (dtype == DDN_F32) hipLaunchKernelGGL(k_reduce_partial<float>, dim3(chunks), dim3(256), 0, st, (const float*)x, (const float*)nullptr, n, chunks, part);
        else hipLaunchKernelGGL(k_reduce_partial<double>, dim3(chunks), dim3(256), 0, st, (const double*)x, (const double*)nullptr, n, chunks, part);
        hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(64), 0, st, part, chunks, 1.0, part + 2 * chunks);
        norm2 = part + 2 * chunks + 2;  // mean d^2 with npix = 1 -> sum of squares
        if (dtype == DDN_F32) hipLaunchKernelGGL(k_prox<float>, dim3(ew_grid(n)), dim3(256), 0, st, kind, (const float*)x, (float*)y, n, (float)lambda, norm2);
        else hipLaunchKernelGGL(k_prox<double>, dim3(ew_grid(n)), dim3(256), 0, st, kind, (const double*)x, (double*)y, n, lambda, norm2);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
        (void)hipFree(part);
        return DDN_OK;
    }
    if (dtype == DDN_F32) hipLaunchKernelGGL(k_prox<float>, dim3(ew_grid(n)), dim3(256), 0, st, kind, (const float*)x, (float*)y, n, (float)lambda, norm2);
    else hipLaunchKernelGGL(k_prox<double>, dim3(ew_grid(n)), dim3(256), 0, st, kind, (const double*)x, (double*)y, n, lambda, norm2);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

int ddn_reduce(int dtype, const void* x, const void* y, size_t n_per_image, int batch, double* out4, void* stream) {
    if (!x || !out4 || batch < 1) return fail(DDN_E_ARG, "reduce: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int chunks = (int)std::min<size_t>(256, std::max<size_t>(1, n_per_image / 4096));
    double* part = nullptr;
    HIPCHK(hipMalloc((void**)&part, sizeof(double) * 2 * (size_t)chunks * batch));
    if (dtype == DDN_F32) hipLaunchKernelGGL(k_reduce_partial<float>, dim3(chunks * batch), dim3(256), 0, st, (const float*)x, (const float*)y, n_per_image, chunks, part);
    else hipLaunchKernelGGL(k_reduce_partial<double>, dim3(chunks * batch), dim3(256), 0, st, (const double*)x, (const double*)y, n_per_image, chunks, part);
    hipLaunchKernelGGL(k_reduce_final, dim3(batch), dim3(64), 0, st, part, chunks, (double)n_per_image, out4);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    (void)hipFree(part);
    return DDN_OK;
}

int ddn_reduce_static(int dtype, const void* x, const void* y, size_t n_per_image, int batch, double* out4, void* stream) {
    if (!x || !out4 || batch < 1 || n_per_image == 0 || (dtype != DDN_F32 && dtype != DDN_F64)) return fail(DDN_E_ARG, "reduce_static: bad arguments");
    if (batch > DDN_REDUCE_STATIC_MAX_BATCH) return fail(DDN_E_UNSUPPORTED, "reduce_static: more than DDN_REDUCE_STATIC_MAX_BATCH images (use ddn_reduce)");
    double* part = nullptr;
    int rc = static_scratch(&part);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    // same launch shape as ddn_reduce, so the two give bit-identical sums; every slot that k_reduce_final
    // reads is written by k_reduce_partial of this call: nothing in the array has to start at zero
    const int chunks = (int)std::min<size_t>(kStaticChunks, std::max<size_t>(1, n_per_image / 4096));
    if (dtype == DDN_F32) hipLaunchKernelGGL(k_reduce_partial<float>, dim3(chunks * batch), dim3(256), 0, st, (const float*)x, (const float*)y, n_per_image, chunks, part);
    else hipLaunchKernelGGL(k_reduce_partial<double>, dim3(chunks * batch), dim3(256), 0, st, (const double*)x, (const double*)y, n_per_image, chunks, part);
    hipLaunchKernelGGL(k_reduce_final, dim3(batch), dim3(64), 0, st, part, chunks, (double)n_per_image, out4);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

int ddn_cp_run(ddn_plan* p, void* x, void* z, void* y, const void* b, double t, double s, double gamma, int objective, int n_iters, double* ring) {
    DDN_WHOLE_PLANE(p, "cp_run");
    if (!p || !x || !z || !y || !b || n_iters < 0) return fail(DDN_E_ARG, "cp_run: bad arguments");
    if (objective != DDN_OBJ_L1 && objective != DDN_OBJ_L2) return fail(DDN_E_ARG, "Invalid objective function. Choose 'l1' or 'l2'.");
    p->ev_row.clear(); p->ev_col.clear(); p->ev_used = 0;
    return p->dtype == DDN_F32 ? cp_run_impl<float>(p, x, z, y, b, t, s, gamma, objective, n_iters, ring)
                               : cp_run_impl<double>(p, x, z, y, b, t, s, gamma, objective, n_iters, ring);
}

int ddn_admm_run(ddn_plan* p, void* u, void* w, void* y, void* z, const void* b, double t, double rho, double gamma, int objective, int n_iters,
                 int tol_kind, double tol, int first_loss_iter, double* ring, int* iters_done, void* sol) {
    DDN_WHOLE_PLANE(p, "admm_run");
    if (!p || !u || !w || !y || !z || !b || n_iters < 0) return fail(DDN_E_ARG, "admm_run: bad arguments");
    if (objective != DDN_OBJ_L1 && objective != DDN_OBJ_L2) return fail(DDN_E_ARG, "Invalid objective function. Choose 'l1' or 'l2'.");
    p->ev_row.clear(); p->ev_col.clear(); p->ev_used = 0;
    return p->dtype == DDN_F32 ? admm_run_impl<float>(p, u, w, y, z, b, t, rho, gamma, objective, n_iters, tol_kind, tol, first_loss_iter, ring, iters_done, sol)
                               : admm_run_impl<double>(p, u, w, y, z, b, t, rho, gamma, objective, n_iters, tol_kind, tol, first_loss_iter, ring, iters_done, sol);
}

int ddn_pdr_run(ddn_plan* p, void* z1, void* z2, const void* b, double t, double rho, double gamma, int objective, int n_iters,
                int tol_kind, double tol, int first_loss_iter, double* ring, int* iters_done) {
    DDN_WHOLE_PLANE(p, "pdr_run");
    if (!p || !z1 || !z2 || !b || n_iters < 0) return fail(DDN_E_ARG, "pdr_run: bad arguments");
    p->ev_row.clear(); p->ev_col.clear(); p->ev_used = 0;
    return p->dtype == DDN_F32 ? pdr_run_impl<float>(p, z1, z2, b, t, rho, gamma, objective, n_iters, tol_kind, tol, first_loss_iter, ring, iters_done)
                               : pdr_run_impl<double>(p, z1, z2, b, t, rho, gamma, objective, n_iters, tol_kind, tol, first_loss_iter, ring, iters_done);
}

int ddn_pddr_run(ddn_plan* p, void* pp, void* q, const void* b, double t, double rho, double gamma, int n_iters,
                 int tol_kind, double tol, int first_loss_iter, double* ring, int* iters_done) {
    DDN_WHOLE_PLANE(p, "pddr_run");
    if (!p || !pp || !q || !b || n_iters < 0) return fail(DDN_E_ARG, "pddr_run: bad arguments");
    p->ev_row.clear(); p->ev_col.clear(); p->ev_used = 0;
    return p->dtype == DDN_F32 ? pddr_run_impl<float>(p, pp, q, b, t, rho, gamma, n_iters, tol_kind, tol, first_loss_iter, ring, iters_done)
                               : pddr_run_impl<double>(p, pp, q, b, t, rho, gamma, n_iters, tol_kind, tol, first_loss_iter, ring, iters_done);
}

// diagnostic: copies the phase stamps of the last fused row kernel launch (16 int64 per workgroup)
int ddn_slab_layout(ddn_plan* p, int* tile_cols, int* n_tiles) {
    if (!p) return fail(DDN_E_ARG, "slab_layout: null plan");
    if (tile_cols) *tile_cols = 1 << p->cs;
    if (n_tiles) *n_tiles = p->ntile;
    return DDN_OK;
}

int ddn_slab_rows(ddn_plan* p, int solver, int rows, void* spec_rows, void* s0, void* s1, void* s2, void* s3, const void* b,
                  double t, double rho, double gamma, int objective, int consume, int produce, void* sol, int sol_only) {
    if (!p || !spec_rows || !s0 || !s1 || !b) return fail(DDN_E_ARG, "slab_rows: null argument");
    if (p->batch != 1) return fail(DDN_E_ARG, "slab_rows: the plan must hold one image");
    if (rows < 1 || rows > p->H) return fail(DDN_E_ARG, "slab_rows: bad row count");
    if (p->tile_pad_rows) return fail(DDN_E_UNSUPPORTED, "slab_rows: padded tiles are not supported");
    if (p->dtype == DDN_F32) return slab_rows_impl<float>(p, solver, rows, spec_rows, s0, s1, s2, s3, b, t, rho, gamma, objective, consume, produce, sol, sol_only);
    return slab_rows_impl<double>(p, solver, rows, spec_rows, s0, s1, s2, s3, b, t, rho, gamma, objective, consume, produce, sol, sol_only);
}

int ddn_slab_cols_part(ddn_plan* p, void* spec_cols, int tile_lo, int n_tiles_local, int n_row_blocks, int mode, int first, int count) {
    if (!p || (!spec_cols && n_tiles_local > 0)) return fail(DDN_E_ARG, "slab_cols: null argument");
    if (p->batch != 1) return fail(DDN_E_ARG, "slab_cols: the plan must hold one image");
    if (tile_lo < 0 || n_tiles_local < 0 || tile_lo + n_tiles_local > p->ntile) return fail(DDN_E_ARG, "slab_cols: bad tile range");
    if (p->tile_pad_rows) return fail(DDN_E_UNSUPPORTED, "slab_cols: padded tiles are not supported");
    if (n_row_blocks < 0) return fail(DDN_E_ARG, "slab_cols: bad number of row blocks");
    if (mode < 0 || mode > 2) return fail(DDN_E_ARG, "slab_cols: mode is 0 (resolvent), 1 (times eK) or 2 (times conj eK)");
    if (mode == 0) {
        if (p->dtype == DDN_F32) return slab_cols_impl<float, COL_RES2>(p, spec_cols, tile_lo, n_tiles_local, 0, n_row_blocks, first, count);
        return slab_cols_impl<double, COL_RES2>(p, spec_cols, tile_lo, n_tiles_local, 0, n_row_blocks, first, count);
    }
    if (p->dtype == DDN_F32) return slab_cols_impl<float, COL_MUL>(p, spec_cols, tile_lo, n_tiles_local, mode == 2 ? 1 : 0, n_row_blocks, first, count);
    return slab_cols_impl<double, COL_MUL>(p, spec_cols, tile_lo, n_tiles_local, mode == 2 ? 1 : 0, n_row_blocks, first, count);
}

int ddn_slab_cols(ddn_plan* p, void* spec_cols, int tile_lo, int n_tiles_local, int n_row_blocks) {
    return ddn_slab_cols_part(p, spec_cols, tile_lo, n_tiles_local, n_row_blocks, 0, 0, n_tiles_local);
}

int ddn_slab_cols_multiply(ddn_plan* p, void* spec_cols, int tile_lo, int n_tiles_local, int conj, int n_row_blocks) {
    return ddn_slab_cols_part(p, spec_cols, tile_lo, n_tiles_local, n_row_blocks, conj ? 2 : 1, 0, n_tiles_local);
}

int ddn_slab_rows_cp(ddn_plan* p, int rows, void* spec_rows, void* spec_aux, void* x, void* z, void* y, const void* b, double t, double s, double gamma,
                     int objective, int step, int consume, int produce) {
    if (!p || !spec_rows || !x || !z || !y || !b) return fail(DDN_E_ARG, "slab_rows_cp: null argument");
    if (objective != DDN_OBJ_L1 && objective != DDN_OBJ_L2) return fail(DDN_E_ARG, "Invalid objective function. Choose 'l1' or 'l2'.");
    if (p->batch != 1) return fail(DDN_E_ARG, "slab_rows_cp: the plan must hold one image");
    if (rows < 1 || rows > p->H) return fail(DDN_E_ARG, "slab_rows_cp: bad row count");
    if (p->tile_pad_rows) return fail(DDN_E_UNSUPPORTED, "slab_rows_cp: padded tiles are not supported");
    if (p->dtype == DDN_F32) return slab_rows_cp_impl<float>(p, rows, spec_rows, spec_aux, x, z, y, b, t, s, gamma, objective, step, consume, produce);
    return slab_rows_cp_impl<double>(p, rows, spec_rows, spec_aux, x, z, y, b, t, s, gamma, objective, step, consume, produce);
}

int ddn_slab_loss(ddn_plan* p, int rows, double* record8) {
    if (!p || !record8) return fail(DDN_E_ARG, "slab_loss: null argument");
    if (p->batch != 1 || rows < 1 || rows > p->H) return fail(DDN_E_ARG, "slab_loss: bad arguments");
    LossFin lf;
    std::memset(&lf, 0, sizeof(lf));
    lf.partials = p->partials;
    lf.ring = record8;
    lf.rows_per_plane = rows;
    lf.iter = 1; lf.first_iter = 1;
    lf.npix = (double)p->H * (double)p->W;
    lf.enabled = 1;
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(256), 0, p->stream, lf);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

int ddn_slab_begin(ddn_plan* p) {
    if (!p) return fail(DDN_E_ARG, "slab_begin: null plan");
    HIPCHK(hipMemsetAsync(p->stop, 0xff, (size_t)p->batch * sizeof(int), p->stream));
    return DDN_OK;
}

int ddn_slab_loss_sums(ddn_plan* p, int rows, double* sums4) {
    if (!p || !sums4) return fail(DDN_E_ARG, "slab_loss_sums: null argument");
    if (p->batch != 1 || rows < 1 || rows > p->H) return fail(DDN_E_ARG, "slab_loss_sums: bad arguments");
    hipLaunchKernelGGL(k_slab_sums, dim3(1), dim3(256), 0, p->stream, (const double*)p->partials, rows, sums4);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

int ddn_slab_stop_test(ddn_plan* p, const double* sums4, int kind, double tol, int iter, int first_iter, double* record8) {
    if (!p || !sums4) return fail(DDN_E_ARG, "slab_stop_test: null argument");
    if (kind < 0 || kind > 4) return fail(DDN_E_ARG, "slab_stop_test: loss kind must be one of DDN_LOSS_NONE .. DDN_LOSS_PSNR");
    hipLaunchKernelGGL(k_slab_stop_test, dim3(1), dim3(1), 0, p->stream, sums4, kind, tol, iter, first_iter, (double)p->H * (double)p->W, p->stop, record8);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

int ddn_slab_iters_done(ddn_plan* p, int* done_dev) {
    if (!p || !done_dev) return fail(DDN_E_ARG, "slab_iters_done: null argument");
    HIPCHK(hipMemcpyAsync(done_dev, p->stop, sizeof(int), hipMemcpyDeviceToDevice, p->stream));
    return DDN_OK;
}

int ddn_plan_debug_stamps(ddn_plan* p, long long* out_host, int n_wg) {
    if (!p || !p->dbg) return fail(DDN_E_STATE, "debug stamps are off (set DDN_DEBUG_STAMPS=1 before plan_create)");
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipMemcpy(out_host, p->dbg, sizeof(long long) * 16 * (size_t)std::min(n_wg, 8192), hipMemcpyDeviceToHost));
    return DDN_OK;
}

int ddn_plan_enable_timing(ddn_plan* p, int on) {
    if (!p) return fail(DDN_E_ARG, "null plan");
    p->timing = on != 0;
    p->timing_stride = on > 1 ? on : 1;
    p->timing_count[0] = p->timing_count[1] = 0;
    return DDN_OK;
}

int ddn_plan_last_timing(ddn_plan* p, double* row_ms, double* col_ms, int* n_row, int* n_col) {
    if (!p) return fail(DDN_E_ARG, "null plan");
    HIPCHK(hipStreamSynchronize(p->stream));
    auto avg = [](std::vector<EvPair>& v, double* out, int* n) {
        double tot = 0;
        for (auto& e : v) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) tot += ms;
        }
        if (out) *out = v.empty() ? 0.0 : tot / (double)v.size();
        if (n) *n = (int)v.size();
    };
    avg(p->ev_row, row_ms, n_row);
    avg(p->ev_col, col_ms, n_col);
    return DDN_OK;
}

}  // extern "C"
