// C-ABI of libomds_hip.so (see include/omds.h for the contract and the reference interfaces
// each entry point replaces).  Host-side orchestration only: packing weights into MFMA fragment
// order, enqueueing the per-horizon-step kernel sequence on the context stream, layout
// conversions for host copies, and the final O(K*n) policy update arithmetic.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include <dlfcn.h>

#include "omds_internal.h"
#ifdef OMDS_TEST_HOOKS
#include "omds_test.h"
#endif

static thread_local std::string g_create_err;

// Optional roctx ranges named like the reference's torch.profiler record_function tags (MPPI.py:102-268,
// frankaPlanner.py:135-145), so that a `rocprofv3 --marker-trace --kernel-trace` timeline reads like the reference's
// Chrome trace.  Enabled with OMDS_ROCTX=1; the roctx library is dlopen'ed, never linked.
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* e = getenv("OMDS_ROCTX");
        if (!e || atoi(e) == 0) return;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
struct RoctxRange {
    static Roctx& api() { static Roctx r; return r; }
    bool on;
    explicit RoctxRange(const char* name) : on(api().push != nullptr) { if (on) api().push(name); }
    ~RoctxRange() { if (on) api().pop(); }
};
}  // namespace

#define CK(expr) OMDS_HIP_CHECK(ctx, expr)
#define REQUIRE(cond, code, msg)            \
    do {                                    \
        if (!(cond)) {                      \
            ctx->err = (msg);               \
            return (code);                  \
        }                                   \
    } while (0)

template <typename T>
static int upload(omds_ctx* ctx, const std::vector<T>& h, const T** dptr) {
    void* p = nullptr;
    CK(hipMalloc(&p, h.size() * sizeof(T)));
    ctx->mlp_allocs.push_back(p);
    CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *dptr = reinterpret_cast<const T*>(p);
    return OMDS_OK;
}

// fp32 -> IEEE binary16 bits, round to nearest even (the screening network's weights, screen_kernel.hip)
static uint16_t f32_to_f16_bits(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0u));   // inf / nan
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                        // rounds to >= 65520: inf
    if (x < 0x38800000u) {                                                                          // subnormal half or zero
        if (x < 0x33000000u) return (uint16_t)sign;                                                 // < 2^-25: zero
        const int shift = 126 - (int)(x >> 23);                                                     // 14 .. 24
        const uint32_t mant = (x & 0x7fffffu) | 0x800000u;
        const uint32_t q = mant >> shift, rem = mant & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        return (uint16_t)(sign | (q + ((rem > halfway || (rem == halfway && (q & 1u))) ? 1u : 0u)));
    }
    const uint32_t e = (x >> 23) - 112u, mant = x & 0x7fffffu;
    uint32_t h = (e << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;                                         // may carry into the exponent: correct
    return (uint16_t)(sign | h);
}

extern "C" {

int omds_version(void) { return 500; }

int omds_device_count(int32_t* count) {
    if (!count) return OMDS_ERR_INVALID_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }   // no device / no driver: zero devices, not a failure
    *count = n;
    return OMDS_OK;
}

void omds_default_params(omds_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->dt = 0.5f;
    p->dst_thr = 0.5f;                                    // MPPI.py:59
    p->lin_thr = 0.015f;                                  // LinDS.py:9
    const float lvel[5] = {0.f, 1.f, -1.f, 0.f, 10.f};    // MPPI.py:132
    const float ln[5] = {0.f, 1.f, 0.f, 0.1f, 100.f};     // MPPI.py:149-153
    const float ltau[5] = {5.f, 1.f, 0.f, 0.1f, 100.f};   // MPPI.py:155 (y_min = ltau_max)
    std::memcpy(p->lvel, lvel, sizeof(lvel));
    std::memcpy(p->ln, ln, sizeof(ln));
    std::memcpy(p->ltau, ltau, sizeof(ltau));
    p->goal_act_cut = 0.5f;
    p->norm_clamp = 0.5f;
    p->coll_slow = 0.1f;
    p->coll_repulse = 0.1f;
    p->softmax_k = -10.f;
    p->rbf_p = 2.f;
    p->ignored_links = 0;
    p->variant = 0;
    p->cost_terms = OMDS_COST_ALL;                        // cost.py:21
}

const char* omds_last_error(const omds_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

static void free_all(omds_ctx* ctx) {
    omds_comm_release(ctx);
    void* ptrs[] = {ctx->d_obs, ctx->d_Fp, ctx->d_radius, ctx->d_trajT, ctx->d_distT, ctx->d_dotT, ctx->d_actT,
                    ctx->d_normalT, ctx->d_kvalT, ctx->d_qdotT, ctx->d_maxact, ctx->d_phisum0, ctx->d_qstage,
                    ctx->d_muT, ctx->d_sigmaT, ctx->d_alphaT, ctx->d_means, ctx->d_Fq, ctx->d_Dmin, ctx->d_idx,
                    ctx->d_gradx, ctx->d_drow, ctx->d_yraw, ctx->d_minidx, ctx->d_dist, ctx->d_nngrad, ctx->d_cost,
                    ctx->d_w, ctx->d_red, ctx->d_stage, ctx->d_cflags, ctx->d_ccounts, ctx->d_coffsets, ctx->d_dscr, ctx->d_A, ctx->d_rowlist, ctx->d_sctotal, ctx->d_scerr, ctx->d_FpH, ctx->d_FqH, ctx->d_evalT, ctx->d_vjp_xyzr, ctx->d_vjp_B, ctx->d_vjp_rad, ctx->d_range, ctx->d_exD, ctx->d_exDr, ctx->d_exMin, ctx->d_exMask, ctx->d_seds, ctx->d_audit_rows, ctx->d_audit_da, ctx->d_FqAll, ctx->d_FqS, ctx->d_FpS, ctx->d_listDa, ctx->d_sinks, ctx->d_qcur, ctx->d_sweepD, ctx->d_sweepDa, ctx->d_sweep_hist, ctx->d_uev, ctx->d_exDeriv, ctx->d_scr_tmp, ctx->d_allDr, ctx->d_allMin, ctx->d_allMask};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (void* p : ctx->mlp_allocs)
        if (p) (void)hipFree(p);
    if (ctx->h_red) (void)hipHostFree(ctx->h_red);
    if (ctx->h_verdict) (void)hipHostFree(ctx->h_verdict);
    if (ctx->h_sinks) (void)hipHostFree(ctx->h_sinks);
    if (ctx->h_in) (void)hipHostFree(ctx->h_in);
    if (ctx->ev_in_q) (void)hipEventDestroy(ctx->ev_in_q);
    if (ctx->ev_in_means) (void)hipEventDestroy(ctx->ev_in_means);
    for (auto e : ctx->prof.start) (void)hipEventDestroy(e);
    for (auto e : ctx->prof.stop) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
}

int omds_create(const omds_config* cfg, omds_ctx** out) {
    if (!cfg || !out) { g_create_err = "omds_create: null argument"; return OMDS_ERR_INVALID_ARG; }
    *out = nullptr;
    if (cfg->n_dof < 1 || cfg->n_dof > OMDS_MAX_DOF || cfg->n_traj < 1 || cfg->horizon < 1 || cfg->n_kernel_max < 1 ||
        cfg->max_obs < 1 || cfg->n_closest < 1 || cfg->n_closest > 64) {
        g_create_err = "omds_create: config out of range (1 <= n_dof <= 7, n_traj, horizon, n_kernel_max, max_obs >= 1, 1 <= n_closest <= 64)";
        return OMDS_ERR_INVALID_ARG;
    }
    if ((long long)cfg->n_traj * cfg->max_obs >= (1LL << 31) || (long long)cfg->n_traj * cfg->n_closest >= (1LL << 31)) {
        g_create_err = "omds_create: n_traj * max_obs (rows of the pair space) must stay below 2^31";
        return OMDS_ERR_INVALID_ARG;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_err = std::string("omds_create: no HIP device available (") + hipGetErrorString(e) +
                       "); this library has no CPU fallback";
        return OMDS_ERR_HIP;
    }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_err = "omds_create: device ordinal out of range"; return OMDS_ERR_INVALID_ARG; }
    omds_ctx* ctx = new (std::nothrow) omds_ctx();
    if (!ctx) { g_create_err = "omds_create: out of host memory"; return OMDS_ERR_INVALID_ARG; }
    ctx->cfg = *cfg;
    ctx->dev = cfg->device;
    omds_default_params(&ctx->prm);
    {   // experiment builds: the audit rate / sweep period of new contexts (the release library has omds_set_screening_audit / _sweep)
        const int v = OMDS_EXP_ENV("OMDS_SCREEN_AUDIT", -1);
        if (v >= 0 && v <= (1 << 20) && (v & (v - 1)) == 0) ctx->audit_one_in = v;
        const int sw = OMDS_EXP_ENV("OMDS_SCREEN_SWEEP", -1);
        if (sw >= 0) ctx->sweep_every = sw;
    }
    auto fail = [&](const std::string& m, int code) {
        g_create_err = m;
        free_all(ctx);
        delete ctx;
        return code;
    };
#define CKC(expr)                                                                         \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(_e), OMDS_ERR_HIP); \
    } while (0)
    CKC(hipSetDevice(ctx->dev));
    CKC(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    const size_t N = cfg->n_traj, H = cfg->horizon, n = cfg->n_dof, Km = cfg->n_kernel_max, Om = cfg->max_obs,
                 k = cfg->n_closest, d = n + 3;
    const size_t rows2 = N * k;
    CKC(hipMalloc(&ctx->d_obs, Om * 4 * 4));
    CKC(hipMalloc(&ctx->d_A, OMDS_MAX_DOF * OMDS_MAX_DOF * 4));
    CKC(hipMalloc(&ctx->d_Fp, std::max(Om, rows2) * OMDS_FROW * 4));   // zeroed by omds_set_mlp (the slot assignment follows the network's d)
    CKC(hipMalloc(&ctx->d_radius, std::max(Om, rows2) * 4));
    CKC(hipMalloc(&ctx->d_FpH, Om * 32 * 2));
    CKC(hipMalloc(&ctx->d_FqH, N * 32 * 2));
    CKC(hipMemsetAsync(ctx->d_FpH, 0, Om * 32 * 2, ctx->stream));
    CKC(hipMemsetAsync(ctx->d_FqH, 0, N * 32 * 2, ctx->stream));
    CKC(hipMalloc(&ctx->d_trajT, H * n * N * 4));
    CKC(hipMalloc(&ctx->d_distT, H * N * 4));
    CKC(hipMalloc(&ctx->d_dotT, H * N * 4));
    CKC(hipMalloc(&ctx->d_actT, H * N * 4));
    CKC(hipMalloc(&ctx->d_normalT, H * n * N * 4));
    CKC(hipMalloc(&ctx->d_kvalT, H * Km * N * 4));
    CKC(hipMalloc(&ctx->d_qdotT, n * N * 4));
    CKC(hipMalloc(&ctx->d_maxact, Km * N * 4));
    CKC(hipMalloc(&ctx->d_phisum0, Km * 4));
    CKC(hipMalloc(&ctx->d_qstage, n * rows2 * 4));
    CKC(hipMalloc(&ctx->d_muT, Km * n * N * 4));
    CKC(hipMalloc(&ctx->d_sigmaT, Km * N * 4));
    CKC(hipMalloc(&ctx->d_alphaT, Km * n * N * 4));
    CKC(hipMalloc(&ctx->d_means, Km * (2 * n + 1) * 4));
    CKC(hipMalloc(&ctx->d_qcur, OMDS_MAX_DOF * 4));
    CKC(hipMalloc(&ctx->d_Fq, rows2 * OMDS_FROW * 4));
    CKC(hipMalloc(&ctx->d_Dmin, N * Om * 4));
    CKC(hipMalloc(&ctx->d_rowlist, N * Om * 4));
    CKC(hipMalloc(&ctx->d_listDa, N * Om * 4));
    CKC(hipMalloc(&ctx->d_range, N * 4 * 4));
    ctx->ex_cap = (int)std::min<size_t>(N * Om, N * 32);   // 32 candidates per rollout on average; longer lists -> fp32 fallback
    CKC(hipMalloc(&ctx->d_exD, (size_t)ctx->ex_cap * 4));
    CKC(hipMalloc(&ctx->d_exDr, (size_t)ctx->ex_cap * 4));
    CKC(hipMalloc(&ctx->d_exMin, (size_t)ctx->ex_cap * 4));
    CKC(hipMalloc(&ctx->d_exMask, (size_t)ctx->ex_cap * (OMDS_MAX_HIDDEN + 1) * 8 * 4));
    CKC(hipMalloc(&ctx->d_sctotal, (H + 2) * 4));
    CKC(hipMalloc(&ctx->d_sinks, H * sizeof(SelectSink)));
    CKC(hipHostMalloc(&ctx->h_sinks, H * sizeof(SelectSink)));
    CKC(hipMalloc(&ctx->d_scerr, 16));
    CKC(hipMalloc(&ctx->d_idx, rows2 * 4));
    CKC(hipMalloc(&ctx->d_gradx, rows2 * d * 4));
    CKC(hipMalloc(&ctx->d_drow, rows2 * 4));
    CKC(hipMalloc(&ctx->d_yraw, rows2 * OMDS_CPAD * 4));
    CKC(hipMalloc(&ctx->d_minidx, rows2 * 4));
    CKC(hipMalloc(&ctx->d_dist, N * 4));
    CKC(hipMalloc(&ctx->d_nngrad, N * n * 4));
    CKC(hipMalloc(&ctx->d_cost, N * 4));
    CKC(hipMalloc(&ctx->d_w, N * 4));
    const size_t redn = std::max<size_t>((size_t)omds_red_size((int)Km, (int)n) + 8, 2 * H + 16);   // also the screening counters of a propagate (4 + 2 (H + 1))
    CKC(hipMalloc(&ctx->d_red, redn * 4));
    CKC(hipHostMalloc(&ctx->h_red, redn * 4));
    CKC(hipHostMalloc(&ctx->h_verdict, (H + 8) * 4));
    CKC(hipHostMalloc(&ctx->h_in, (Km * (2 * n + 1) + OMDS_MAX_DOF) * 4));   // pinned staging of the small per-iteration inputs
    CKC(hipEventCreateWithFlags(&ctx->ev_in_q, hipEventDisableTiming));
    CKC(hipEventCreateWithFlags(&ctx->ev_in_means, hipEventDisableTiming));
    ctx->stage_bytes = std::max({N * H * std::max(Km, n) * 4, N * Om * 4, Km * n * N * 4, rows2 * OMDS_CPAD * 4});
    CKC(hipMalloc(&ctx->d_stage, ctx->stage_bytes));
    CKC(hipMalloc(&ctx->d_cflags, N * H));
    CKC(hipMalloc(&ctx->d_ccounts, N * 4));
    CKC(hipMalloc(&ctx->d_coffsets, (N + 1) * 4));
    CKC(hipMemsetAsync(ctx->d_trajT, 0, H * n * N * 4, ctx->stream));
    CKC(hipMemsetAsync(ctx->d_kvalT, 0, H * Km * N * 4, ctx->stream));
    CKC(hipMemsetAsync(ctx->d_maxact, 0, Km * N * 4, ctx->stream));
    CKC(hipMemsetAsync(ctx->d_phisum0, 0, Km * 4, ctx->stream));
    CKC(hipStreamSynchronize(ctx->stream));
#undef CKC
    *out = ctx;
    return OMDS_OK;
}

void omds_destroy(omds_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->dev);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_all(ctx);
    delete ctx;
}

int omds_sync(omds_ctx* ctx) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    CK(hipStreamSynchronize(ctx->stream));
    return OMDS_OK;
}

#ifdef OMDS_TIMELINE
// Diagnostic build only (not in include/omds.h): phase timestamps of the last k_pass1 launch, see tools/pass1_timeline.py.
extern "C" OMDS_API int omds_timeline_fetch(omds_ctx* ctx, unsigned long long* host, int n_workgroups) {
    if (!ctx || !ctx->mlp.tl || n_workgroups > (1 << 16)) return OMDS_ERR_INVALID_ARG;
    CK(hipStreamSynchronize(ctx->stream));
    CK(hipMemcpy(host, ctx->mlp.tl, (size_t)n_workgroups * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return OMDS_OK;
}
#endif

// ---- weights -------------------------------------------------------------------------------------

int omds_set_mlp(omds_ctx* ctx, int n_linear, const int32_t* dims, const float* const* W, const float* const* b, int act,
                 float out_div) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(dims && n_linear >= 2, OMDS_ERR_INVALID_ARG, "omds_set_mlp: null argument or fewer than 2 Linear layers");
    return omds_set_mlp_ex(ctx, n_linear, dims, dims + 1, W, b, act, out_div, 0, nullptr);
}

// Everything omds_set_mlp_ex derives from the caller's weights on the HOST: validation, zero-padding to the kernels' width, and the
// MFMA fragment packs.  No device, no context: the sanitizer build runs it on the CPU (tests/test_asan_cpu.py through
// omds_test_pack_mlp).
struct MlpPacks {
    int nhh = 0, C = 0, d = 0, act = 0;
    float out_div = 1.f;
    uint32_t skip_mask = 0;
    uint8_t skip_col[OMDS_MAX_HIDDEN + 1] = {0};
    std::vector<float4> wf, wb, wf16, wb16, wb4, wf4, wl, w1b, w1b16, w1f, w1f16;
    std::vector<float> bh, bl, wlraw, whraw, w1t, b1, sbias, wht, wlt;
    std::vector<uint16_t> wh;
    double f_fwd = 0.0, f_bwd = 0.0;
    // what build_screen_pack needs to build wh / sbias again in another unit order (ReLU / tanh networks the screening kernel takes)
    std::vector<std::vector<float>> host_W, host_b;   // zero-padded to width 256
    std::vector<int32_t> out_dims;
};

// fp16 screening network (screen_kernel.hip): slices of 32 output rows x 256 k in A-fragment order of v_mfma_f32_32x32x16_f16, with
// the k order permuted to the C layout of the previous layer (chunk cc, lane-half h, slot j <-> position 16cc + 8(j>>2) + 4h + (j&3)).
// Behind a skip concatenation (level L = output of Linear L) the consuming layer's input columns are packed in a VIRTUAL
// order: its own c0 = out_dims[L] columns first, the 3d concatenated input columns LAST (virtual 256 - 3d .. 255 = k-chunks
// 14 and 15 whatever c0 is -- the K order of a dot product is free), zeros in between: omds_screen_sidx puts the inputs there.
// order (optional, [nhh + 1][256]): order[L][p] = the hidden unit of Linear L that sits at POSITION p of the screening network --
// row p of that layer's slices, k position p of the layer behind it.  Any permutation computes the same function; the kernel
// skips k-chunks whose 16 units are zero for all pairs of a wave, so the units that seldom or never fire are put together
// (screen_reorder).  nullptr: the identity.
static void build_screen_pack(MlpPacks& pk, const int32_t* order) {
    const int nhh = pk.nhh, C = pk.C, Wd = OMDS_WIDTH, F = 3 * pk.d, n_linear = nhh + 2;
    const uint32_t skip_mask = pk.skip_mask;
    std::vector<const float*> Wv(n_linear), bv(n_linear);
    for (int i = 0; i < n_linear; ++i) { Wv[i] = pk.host_W[i].data(); bv[i] = pk.host_b[i].data(); }
    const float* const* W = Wv.data();
    const float* const* b = bv.data();
    auto unit = [&](int L, int p) -> int { return order ? order[(size_t)L * Wd + p] : p; };   // position p of Linear L's outputs
    auto real_col = [&](int consumer, int v) -> int {   // virtual input position v of Linear `consumer` -> column of Wpad, -1 = zero
        const int L = consumer - 1;
        if (L < 0 || !((skip_mask >> L) & 1u)) return unit(L, v);
        const int c0 = pk.out_dims[L];
        if (v < c0) return v;
        if (v >= Wd - F) return c0 + (v - (Wd - F));
        return -1;
    };
    // tanh: every layer in front of an activation is scaled by 2 log2(e) (screen_kernel.hip: act_pk); the last layer is not
    const float hs = pk.act == OMDS_ACT_TANH ? OMDS_SCREEN_TANH_SCALE : 1.f;
    std::vector<uint16_t>& wh = pk.wh;
    const int nsl = nhh * 8 + 2;
    wh.assign((size_t)nsl * 16 * 64 * 8, 0);
    pk.sbias.assign((size_t)(nhh + 2) * Wd, 0.f);
    // slice 0: layer 1, fragment 2 fb + cc = positions 32 fb .. +31 x inputs 16 cc .. +15 (slot j of lane-half h = input 16cc + 8h + j)
    for (int fb = 0; fb < 8; ++fb)
        for (int cc = 0; cc < 2; ++cc)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int r = unit(0, 32 * fb + (lane & 31)), kk = 16 * cc + 8 * (lane >> 5) + j;
                    const float v = (kk < F) ? W[0][(size_t)r * F + kk] : 0.f;
                    wh[(((size_t)(2 * fb + cc)) * 64 + lane) * 8 + j] = f32_to_f16_bits(hs * v);
                }
    for (int sl = 1; sl < nsl; ++sl) {
        const bool lastl = sl == nsl - 1;
        const int lin = lastl ? n_linear - 1 : (sl - 1) / 8 + 1;   // the Linear layer this slice belongs to
        const float* Wsrc = W[lin];
        const int fb = lastl ? 0 : (sl - 1) % 8, rows = lastl ? C : Wd;
        for (int cc = 0; cc < 16; ++cc)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int rp = 32 * fb + (lane & 31), r = (lastl || rp >= Wd) ? rp : unit(lin, rp);
                    const int kk = real_col(lin, 16 * cc + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3));
                    const float v = (rp < rows && kk >= 0) ? Wsrc[(size_t)r * Wd + kk] : 0.f;
                    wh[(((size_t)sl * 16 + cc) * 64 + lane) * 8 + j] = f32_to_f16_bits(lastl ? v : hs * v);
                }
    }
    for (int L = 0; L <= nhh; ++L)
        for (int p = 0; p < Wd; ++p) pk.sbias[(size_t)L * Wd + p] = hs * b[L][unit(L, p)];
    std::memcpy(&pk.sbias[(size_t)(nhh + 1) * Wd], b[n_linear - 1], C * sizeof(float));
}
#define PREQ(cond, code, msg) do { if (!(cond)) { err = (msg); return (code); } } while (0)
static int build_mlp_packs(int n, int n_linear, const int32_t* in_dims, const int32_t* out_dims, const float* const* W,
                           const float* const* b, int act, float out_div, int n_skips, const int32_t* skip_after, MlpPacks& pk,
                           std::string& err) {
    PREQ(in_dims && out_dims && W && b && n_linear >= 2, OMDS_ERR_INVALID_ARG, "omds_set_mlp: null argument or fewer than 2 Linear layers");
    PREQ(n_skips == 0 || skip_after, OMDS_ERR_INVALID_ARG, "omds_set_mlp_ex: n_skips > 0 needs skip_after");
    PREQ(in_dims[0] == 3 * (n + 3) || in_dims[0] == 3 * (n + 2), OMDS_ERR_INVALID_ARG,
         "omds_set_mlp: dims[0] must be 3*(n_dof+3), or 3*(n_dof+2) for planar obstacle points (NeRF encoding [x, sin x, cos x])");
    const int d = in_dims[0] / 3;
    PREQ(3 * d <= 32, OMDS_ERR_UNSUPPORTED, "omds_set_mlp: 3*(n_dof+3) > 32 not supported");
    const int nhid = n_linear - 1;
    PREQ(nhid <= OMDS_MAX_HIDDEN, OMDS_ERR_UNSUPPORTED, "omds_set_mlp: too many hidden layers");
    uint32_t skip_mask = 0;   // bit i: the encoded input is concatenated behind the activations of Linear i (network_macros_mod.py:142-146)
    for (int s = 0; s < n_skips; ++s) {
        PREQ(skip_after[s] >= 0 && skip_after[s] < nhid, OMDS_ERR_INVALID_ARG,
             "omds_set_mlp_ex: skip_after entries must name a hidden Linear layer (0 .. n_linear-2)");
        skip_mask |= 1u << skip_after[s];
    }
    for (int i = 0; i < nhid; ++i)
        PREQ(out_dims[i] >= 1 && out_dims[i] + (((skip_mask >> i) & 1u) ? 3 * d : 0) <= OMDS_WIDTH, OMDS_ERR_UNSUPPORTED,
             "omds_set_mlp: hidden widths (plus a concatenated input) above 256 are not supported by the MFMA kernels (narrower layers are zero-padded to width 256)");
    for (int i = 1; i < n_linear; ++i)
        PREQ(in_dims[i] == out_dims[i - 1] + (((skip_mask >> (i - 1)) & 1u) ? 3 * d : 0), OMDS_ERR_INVALID_ARG,
             "omds_set_mlp: the input width of a Linear layer must be the previous output width (+ 3*(n_dof+3) behind a skip concatenation)");
    const int C = out_dims[n_linear - 1];
    PREQ(C >= 1 && C <= OMDS_CPAD, OMDS_ERR_UNSUPPORTED, "omds_set_mlp: 1 <= out_channels <= 16 required");
    PREQ(act == OMDS_ACT_RELU || act == OMDS_ACT_TANH, OMDS_ERR_UNSUPPORTED, "omds_set_mlp: act must be OMDS_ACT_RELU or OMDS_ACT_TANH");
    PREQ(out_div != 0.f, OMDS_ERR_INVALID_ARG, "omds_set_mlp: out_div must be non-zero");
    for (int i = 0; i < n_linear; ++i) PREQ(W[i] && b[i], OMDS_ERR_INVALID_ARG, "omds_set_mlp: null weight or bias array");
    // Narrower hidden layers (the reference also ships 128-wide nets) are zero-padded to the kernels' width:
    // padded units have zero weights and biases on both sides, so relu/tanh(0) = 0 feeds nothing forward and
    // receives no gradient -- outputs and gradients are unchanged (the padded MFMA work is wasted, not wrong).
    // A concatenated input keeps its place: its columns follow the (narrower) layer's own outputs in the padded row.
    std::vector<std::vector<float>> Wpad(n_linear), bpad(n_linear);
    std::vector<const float*> Wp(n_linear), bp(n_linear);
    for (int i = 0; i < n_linear; ++i) {
        const int in = in_dims[i], out = out_dims[i];
        const int pin = i == 0 ? in : OMDS_WIDTH, pout = i == n_linear - 1 ? out : OMDS_WIDTH;
        Wpad[i].assign((size_t)pout * pin, 0.f);
        bpad[i].assign((size_t)pout, 0.f);
        for (int o = 0; o < out; ++o) {
            std::memcpy(&Wpad[i][(size_t)o * pin], &W[i][(size_t)o * in], (size_t)in * sizeof(float));
            bpad[i][o] = b[i][o];
        }
        Wp[i] = Wpad[i].data();
        bp[i] = bpad[i].data();
    }
    W = Wp.data();
    b = bp.data();
    pk.nhh = nhid - 1;
    pk.C = C;
    pk.d = d;
    pk.out_div = out_div;
    pk.act = act;
    pk.skip_mask = skip_mask;
    for (int i = 0; i < nhid; ++i) pk.skip_col[i] = (uint8_t)(((skip_mask >> i) & 1u) ? out_dims[i] : 0);
    const int nhh = pk.nhh;
    const int Wd = OMDS_WIDTH;
    // hidden->hidden: forward and transposed (backward) fragment packs
    std::vector<float4>&wf = pk.wf, &wb = pk.wb;
    wf.assign((size_t)std::max(nhh, 1) * OMDS_NCB * 32 * 64, make_float4(0, 0, 0, 0));
    wb.assign(wf.size(), make_float4(0, 0, 0, 0));
    pk.bh.assign((size_t)std::max(nhh, 1) * Wd, 0.f);
    for (int l = 0; l < nhh; ++l) {
        const float* Wl = W[l + 1];
        for (int cb = 0; cb < OMDS_NCB; ++cb)
            for (int c = 0; c < 32; ++c)
                for (int lane = 0; lane < 64; ++lane) {
                    // the lane's fragment covers POSITIONS 8c + 4(lane>>5) .. +3 of the k-permuted tile: columns omds_kat(position)
                    const int j = 32 * cb + (lane & 31), s0 = 8 * c + 4 * (lane >> 5);
                    const int k0 = omds_kat(s0), k1 = omds_kat(s0 + 1), k2 = omds_kat(s0 + 2), k3 = omds_kat(s0 + 3);
                    const size_t o = (((size_t)l * OMDS_NCB + cb) * 32 + c) * 64 + lane;
                    wf[o] = make_float4(Wl[j * Wd + k0], Wl[j * Wd + k1], Wl[j * Wd + k2], Wl[j * Wd + k3]);
                    wb[o] = make_float4(Wl[k0 * Wd + j], Wl[k1 * Wd + j], Wl[k2 * Wd + j], Wl[k3 * Wd + j]);
                }
        std::memcpy(&pk.bh[(size_t)l * Wd], b[l + 1], Wd * sizeof(float));
    }
    // 16-row packs (v_mfma_f32_16x16x4): the four tile positions of lane group g in steps 0..3 of chunk c are
    // 16c + pa[g] + {0, 2, 8, 10}, pa = {0, 4, 1, 5} -- the position SEQUENCE of the 32-row kernels (8c'+{0,4,1,5,2,6,3,7}), i.e.
    // columns 16c + 0 .. 15 in ascending order (omds_kat), so a 16-row tile is bit-identical to a 32-row tile (both MFMAs are
    // fmaf chains in k order, tools/ubench/mfma_order.hip)
    auto k16 = [](int c, int g, int mm) { return omds_kat(16 * c + ((g >> 1) + 4 * (g & 1)) + 8 * (mm >> 1) + 2 * (mm & 1)); };
    std::vector<float4>&wf16 = pk.wf16, &wb16 = pk.wb16;
    wf16.assign(wf.size(), make_float4(0, 0, 0, 0));
    wb16.assign(wf.size(), make_float4(0, 0, 0, 0));
    for (int l = 0; l < nhh; ++l) {
        const float* Wl = W[l + 1];
        for (int cb = 0; cb < 16; ++cb)
            for (int c = 0; c < 16; ++c)
                for (int lane = 0; lane < 64; ++lane) {
                    const int j = 16 * cb + (lane & 15), g = lane >> 4;
                    const int k0 = k16(c, g, 0), k1 = k16(c, g, 1), k2 = k16(c, g, 2), k3 = k16(c, g, 3);
                    const size_t o = (((size_t)l * 16 + cb) * 16 + c) * 64 + lane;
                    wf16[o] = make_float4(Wl[j * Wd + k0], Wl[j * Wd + k1], Wl[j * Wd + k2], Wl[j * Wd + k3]);
                    wb16[o] = make_float4(Wl[k0 * Wd + j], Wl[k1 * Wd + j], Wl[k2 * Wd + j], Wl[k3 * Wd + j]);
                }
    }
    // backward pack for the 4-row-group GEMM (v_mfma_f32_4x4x1, mlp_device.h gemm4): the gradient at a layer's inputs is
    // sum_k G[row][k] W[k][j] (W [out = k][in = j]); lane l of column block cb holds W[4 kq .. 4 kq + 3][64 cb + l]
    // (and the forward pack of the same GEMM, pass2_body_g4: sum_k H[row][k] W[j][k], lane l of block cb holds W[64 cb + l][k0 .. k3])
    std::vector<float4>&wb4 = pk.wb4, &wf4 = pk.wf4;
    wb4.assign((size_t)std::max(nhh, 1) * 4 * 64 * 64, make_float4(0, 0, 0, 0));
    wf4.assign(wb4.size(), make_float4(0, 0, 0, 0));
    for (int l = 0; l < nhh; ++l) {
        const float* Wl = W[l + 1];
        for (int cb = 0; cb < 4; ++cb)
            for (int kq = 0; kq < 64; ++kq)
                for (int lane = 0; lane < 64; ++lane) {
                    const int j = 64 * cb + lane, s0 = 4 * kq;   // positions 4 kq .. +3 of the gradient tile
                    const int k0 = omds_kat(s0), k1 = omds_kat(s0 + 1), k2 = omds_kat(s0 + 2), k3 = omds_kat(s0 + 3);
                    wb4[(((size_t)l * 4 + cb) * 64 + kq) * 64 + lane] =
                        make_float4(Wl[(size_t)k0 * Wd + j], Wl[(size_t)k1 * Wd + j], Wl[(size_t)k2 * Wd + j], Wl[(size_t)k3 * Wd + j]);
                    wf4[(((size_t)l * 4 + cb) * 64 + kq) * 64 + lane] =
                        make_float4(Wl[(size_t)j * Wd + k0], Wl[(size_t)j * Wd + k1], Wl[(size_t)j * Wd + k2], Wl[(size_t)j * Wd + k3]);
                }
    }
    // last layer: 16x16x4 B-fragments, channels padded to 16
    const float* WL = W[n_linear - 1];
    std::vector<float4>& wl = pk.wl;
    wl.assign(16 * 64, make_float4(0, 0, 0, 0));
    for (int c = 0; c < 16; ++c)
        for (int lane = 0; lane < 64; ++lane) {
            const int j = lane & 15;   // the last layer reads the tile like gemm16 (load_a16): the same ascending k sequence
            float v[4] = {0, 0, 0, 0};
            if (j < C)
                for (int mm = 0; mm < 4; ++mm) v[mm] = WL[j * Wd + k16(c, lane >> 4, mm)];
            wl[c * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
        }
    pk.bl.assign(OMDS_CPAD, 0.f);
    pk.wlraw.assign((size_t)C * Wd, 0.f);
    std::memcpy(pk.bl.data(), b[n_linear - 1], C * sizeof(float));
    std::memcpy(pk.wlraw.data(), WL, (size_t)C * Wd * sizeof(float));
    // first layer: transposed copy + backward pack over the 3d features (padded to 32 columns)
    const int F = 3 * d;
    pk.w1t.assign((size_t)F * Wd, 0.f);
    pk.b1.assign(Wd, 0.f);
    for (int c = 0; c < Wd; ++c)
        for (int f = 0; f < F; ++f) pk.w1t[(size_t)f * Wd + c] = W[0][c * F + f];
    std::memcpy(pk.b1.data(), b[0], Wd * sizeof(float));
    std::vector<float4>& w1b = pk.w1b;
    w1b.assign(32 * 64, make_float4(0, 0, 0, 0));
    for (int c = 0; c < 32; ++c)
        for (int lane = 0; lane < 64; ++lane) {
            const int f = lane & 31, s0 = 8 * c + 4 * (lane >> 5);
            float v[4] = {0, 0, 0, 0};
            if (f < F)
                for (int mm = 0; mm < 4; ++mm) v[mm] = W[0][omds_kat(s0 + mm) * F + f];
            w1b[c * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
        }
    std::vector<float4>& w1b16 = pk.w1b16;
    w1b16.assign(16 * 2 * 64, make_float4(0, 0, 0, 0));
    for (int c = 0; c < 16; ++c)
        for (int jb = 0; jb < 2; ++jb)
            for (int lane = 0; lane < 64; ++lane) {
                const int f = 16 * jb + (lane & 15), g = lane >> 4;
                float v[4] = {0, 0, 0, 0};
                if (f < F)
                    for (int mm = 0; mm < 4; ++mm) v[mm] = W[0][k16(c, g, mm) * F + f];
                w1b16[(c * 2 + jb) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
            }
    // first layer, forward: K = 32 over the encoded inputs at positions 0..31 of the tile rows (feature omds_kat(position), zero
    // weights for the padding 3d .. 31), in the fragment orders of gemm_k32 and gemm16_k32
    std::vector<float4>& w1f = pk.w1f;
    w1f.assign(OMDS_NCB * 4 * 64, make_float4(0, 0, 0, 0));
    for (int cb = 0; cb < OMDS_NCB; ++cb)
        for (int c = 0; c < 4; ++c)
            for (int lane = 0; lane < 64; ++lane) {
                const int j = 32 * cb + (lane & 31), s0 = 8 * c + 4 * (lane >> 5);
                float v[4];
                for (int mm = 0; mm < 4; ++mm) { const int kk = omds_kat(s0 + mm); v[mm] = kk < F ? W[0][(size_t)j * F + kk] : 0.f; }
                w1f[((size_t)cb * 4 + c) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
            }
    std::vector<float4>& w1f16 = pk.w1f16;
    w1f16.assign(16 * 2 * 64, make_float4(0, 0, 0, 0));
    for (int cb = 0; cb < 16; ++cb)
        for (int c = 0; c < 2; ++c)
            for (int lane = 0; lane < 64; ++lane) {
                const int j = 16 * cb + (lane & 15), g = lane >> 4;
                float v[4];
                for (int mm = 0; mm < 4; ++mm) { const int kk = k16(c, g, mm); v[mm] = kk < F ? W[0][(size_t)j * F + kk] : 0.f; }
                w1f16[((size_t)cb * 2 + c) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
            }
    pk.whraw.assign((size_t)std::max(nhh, 1) * Wd * Wd, 0.f);
    for (int l = 0; l < nhh; ++l) std::memcpy(&pk.whraw[(size_t)l * Wd * Wd], W[l + 1], (size_t)Wd * Wd * sizeof(float));
    if ((act == OMDS_ACT_RELU || act == OMDS_ACT_TANH) && nhh >= 1 && nhh <= 4) {
        // the padded fp32 weights stay on the host: the screening pack is built again when the unit order changes (screen_reorder)
        pk.host_W.assign(Wpad.begin(), Wpad.end());
        pk.host_b.assign(bpad.begin(), bpad.end());
        pk.out_dims.assign(out_dims, out_dims + n_linear);
        build_screen_pack(pk, nullptr);
    }
    // transposed copies for the per-tile compaction (pass1_tile_dyn): row k = the weights leaving unit k
    pk.wht.assign((size_t)std::max(nhh, 1) * Wd * Wd, 0.f);
    for (int l = 0; l < nhh; ++l)
        for (int j = 0; j < Wd; ++j)
            for (int kk = 0; kk < Wd; ++kk) pk.wht[((size_t)l * Wd + kk) * Wd + j] = W[l + 1][(size_t)j * Wd + kk];
    pk.wlt.assign((size_t)Wd * 16, 0.f);
    for (int j = 0; j < C; ++j)
        for (int kk = 0; kk < Wd; ++kk) pk.wlt[(size_t)kk * 16 + j] = WL[(size_t)j * Wd + kk];
    pk.f_fwd = 0.0;
    for (int i = 0; i < n_linear; ++i) pk.f_fwd += 2.0 * in_dims[i] * out_dims[i];   // algorithmic: un-padded
    pk.f_bwd = pk.f_fwd - 2.0 * in_dims[n_linear - 1] * out_dims[n_linear - 1];   // no weight-gradient, no last-layer GEMM
    return OMDS_OK;
}
#undef PREQ

// A network with a hidden layer wider than the fused kernels' 256 columns (MLPRegression is width-agnostic,
// network_macros_mod.py:96-135): raw weights on the device and the buffers of the unfused GEMM path (wide_kernels.hip).
static int set_mlp_wide(omds_ctx* ctx, int n_linear, const int32_t* in_dims, const int32_t* out_dims, const float* const* W,
                        const float* const* b, int act, float out_div, int n_skips) {
    const int n = ctx->cfg.n_dof;
    REQUIRE(n_skips == 0, OMDS_ERR_UNSUPPORTED, "omds_set_mlp_ex: skip concatenations are supported for hidden widths <= 256 only");
    REQUIRE(in_dims[0] == 3 * (n + 3) || in_dims[0] == 3 * (n + 2), OMDS_ERR_INVALID_ARG,
            "omds_set_mlp: dims[0] must be 3*(n_dof+3), or 3*(n_dof+2) for planar obstacle points (NeRF encoding [x, sin x, cos x])");
    const int d = in_dims[0] / 3, C = out_dims[n_linear - 1];
    REQUIRE(3 * d <= 32, OMDS_ERR_UNSUPPORTED, "omds_set_mlp: 3*(n_dof+3) > 32 not supported");
    REQUIRE(n_linear - 1 <= OMDS_MAX_HIDDEN, OMDS_ERR_UNSUPPORTED, "omds_set_mlp: too many hidden layers");
    REQUIRE(C >= 1 && C <= OMDS_CPAD, OMDS_ERR_UNSUPPORTED, "omds_set_mlp: 1 <= out_channels <= 16 required");
    REQUIRE(act == OMDS_ACT_RELU || act == OMDS_ACT_TANH, OMDS_ERR_UNSUPPORTED, "omds_set_mlp: act must be OMDS_ACT_RELU or OMDS_ACT_TANH");
    REQUIRE(out_div != 0.f, OMDS_ERR_INVALID_ARG, "omds_set_mlp: out_div must be non-zero");
    int wmax = in_dims[0];
    for (int i = 0; i < n_linear; ++i) {
        REQUIRE(W[i] && b[i], OMDS_ERR_INVALID_ARG, "omds_set_mlp: null weight or bias array");
        REQUIRE(out_dims[i] >= 1 && out_dims[i] <= 4096, OMDS_ERR_UNSUPPORTED, "omds_set_mlp: layer widths above 4096 are not supported");
        REQUIRE(i == 0 || in_dims[i] == out_dims[i - 1], OMDS_ERR_INVALID_ARG,
                "omds_set_mlp: the input width of a Linear layer must be the previous output width");
        wmax = std::max(wmax, (int)out_dims[i]);
    }
    CK(hipSetDevice(ctx->dev));
    CK(hipStreamSynchronize(ctx->stream));
    for (void* p : ctx->mlp_allocs) (void)hipFree(p);
    ctx->mlp_allocs.clear();
    ctx->have_mlp = false;
    ctx->wide = WideNet{};   // a previous wide network's buffers were in mlp_allocs
    ctx->screen = ScreenDev{};
    ctx->screen_ok = false;
    ctx->screen_cal = false;
    ctx->screen_suspended = false;
    if (ctx->d_dscr) { (void)hipFree(ctx->d_dscr); ctx->d_dscr = nullptr; }
    if (ctx->d_exDeriv) { (void)hipFree(ctx->d_exDeriv); ctx->d_exDeriv = nullptr; }
    WideNet w;
    w.on = true;
    w.d = d; w.act = act; w.out_div = out_div;
    w.dims.assign(1, in_dims[0]);
    for (int i = 0; i < n_linear; ++i) w.dims.push_back(out_dims[i]);
    auto dalloc = [&](float** p, size_t floats) -> int {
        void* q = nullptr;
        CK(hipMalloc(&q, floats * sizeof(float)));
        ctx->mlp_allocs.push_back(q);
        *p = static_cast<float*>(q);
        return OMDS_OK;
    };
    int rc;
    for (int i = 0; i < n_linear; ++i) {
        float *dw = nullptr, *db = nullptr;
        const size_t nw = (size_t)in_dims[i] * out_dims[i];
        if ((rc = dalloc(&dw, nw)) || (rc = dalloc(&db, out_dims[i]))) return rc;
        CK(hipMemcpy(dw, W[i], nw * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, b[i], (size_t)out_dims[i] * 4, hipMemcpyHostToDevice));
        w.W.push_back(dw);
        w.b.push_back(db);
    }
    // pass 1 in chunks of ~256 MB per activation buffer; pass 2 keeps every layer's activation of its n_traj * n_closest rows
    const long long pairs = (long long)ctx->cfg.n_traj * ctx->cfg.max_obs;
    w.chunk_rows = (int)std::min<long long>(pairs, std::max<long long>(8192, std::min<long long>(262144, (1LL << 26) / wmax)));
    w.rows2 = ctx->cfg.n_traj * ctx->cfg.n_closest;
    if ((rc = dalloc(&w.X, (size_t)w.chunk_rows * in_dims[0])) || (rc = dalloc(&w.H[0], (size_t)w.chunk_rows * wmax)) ||
        (rc = dalloc(&w.H[1], (size_t)w.chunk_rows * wmax)) || (rc = dalloc(&w.X2, (size_t)w.rows2 * in_dims[0])) ||
        (rc = dalloc(&w.G[0], (size_t)w.rows2 * wmax)) || (rc = dalloc(&w.G[1], (size_t)w.rows2 * wmax)))
        return rc;
    for (int i = 0; i < n_linear; ++i) {
        float* a = nullptr;
        if ((rc = dalloc(&a, (size_t)w.rows2 * out_dims[i]))) return rc;
        w.A.push_back(a);
    }
    MlpDev m{};   // the fields the stand-alone kernels around the network read (k_modulate, k_blend, the cost)
    m.nhh = n_linear - 2; m.C = C; m.d = d; m.n_dof = n; m.out_div = out_div; m.act = act;
    ctx->mlp = m;
    ctx->wide = w;
    ctx->act = act;
    ctx->f_fwd = 0.0;
    for (int i = 0; i < n_linear; ++i) ctx->f_fwd += 2.0 * in_dims[i] * out_dims[i];
    ctx->f_bwd = ctx->f_fwd - 2.0 * in_dims[n_linear - 1] * out_dims[n_linear - 1];
    ctx->have_mlp = true;
    return OMDS_OK;
}

int omds_set_mlp_ex(omds_ctx* ctx, int n_linear, const int32_t* in_dims, const int32_t* out_dims, const float* const* W,
                    const float* const* b, int act, float out_div, int n_skips, const int32_t* skip_after) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    const int n = ctx->cfg.n_dof;
    if (in_dims && out_dims && W && b && n_linear >= 2 && n_linear <= OMDS_MAX_HIDDEN + 1) {
        bool is_wide = false;
        for (int i = 0; i + 1 < n_linear; ++i) is_wide = is_wide || out_dims[i] > OMDS_WIDTH;
        if (is_wide) return set_mlp_wide(ctx, n_linear, in_dims, out_dims, W, b, act, out_div, n_skips);
    }
    // everything that can reject the call happens before the context is touched: a rejected network (bad act, bad dims, null W)
    // leaves the previous one -- fused or wide -- installed and usable
    MlpPacks pk;
    int rc;
    if ((rc = build_mlp_packs(n, n_linear, in_dims, out_dims, W, b, act, out_div, n_skips, skip_after, pk, ctx->err))) return rc;
    const int nhid = n_linear - 1;
    const uint32_t skip_mask = pk.skip_mask;
    CK(hipSetDevice(ctx->dev));
    CK(hipStreamSynchronize(ctx->stream));
    for (void* p : ctx->mlp_allocs) (void)hipFree(p);
    ctx->mlp_allocs.clear();
    ctx->have_mlp = false;
    ctx->wide = WideNet{};   // its buffers were in mlp_allocs
    MlpDev m{};
    m.nhh = pk.nhh;
    m.C = pk.C;
    m.d = pk.d;
    m.n_dof = n;
    m.out_div = out_div;
    m.act = act;
    m.skip_mask = skip_mask;
    std::memcpy(m.skip_col, pk.skip_col, sizeof(m.skip_col));
    {   // the encoded-input tables keep zeros in the slots the other operand owns and in the padding; the slot assignment follows d
        const size_t rows2 = (size_t)ctx->cfg.n_traj * ctx->cfg.n_closest, rowsB = std::max((size_t)ctx->cfg.max_obs, rows2);
        CK(hipMemsetAsync(ctx->d_Fq, 0, rows2 * OMDS_FROW * 4, ctx->stream));
        CK(hipMemsetAsync(ctx->d_Fp, 0, rowsB * OMDS_FROW * 4, ctx->stream));
        if (ctx->d_FqAll) CK(hipMemsetAsync(ctx->d_FqAll, 0, (size_t)ctx->cfg.n_traj * ctx->cfg.horizon * OMDS_FROW * 4, ctx->stream));
        if (ctx->d_vjp_B) CK(hipMemsetAsync(ctx->d_vjp_B, 0, (size_t)ctx->vjp_cap * OMDS_FROW * 4, ctx->stream));
    }
    if (ctx->d_dscr) { (void)hipFree(ctx->d_dscr); ctx->d_dscr = nullptr; }
    if (ctx->d_exDeriv) { (void)hipFree(ctx->d_exDeriv); ctx->d_exDeriv = nullptr; }
    if (act == OMDS_ACT_TANH) {   // pass 2 keeps 1 - h^2 of every hidden layer for the backward (ReLU uses LDS bit masks)
        const size_t rows = std::max(((size_t)ctx->cfg.n_traj * ctx->cfg.n_closest + 31) / 32 * 32,
                                     (size_t)omds_tail_scratch_rows(ctx->cfg.n_traj, ctx->cfg.n_closest));
        CK(hipMalloc(&ctx->d_dscr, (size_t)nhid * rows * OMDS_WIDTH * 4));
        // the screened step's hand-over: the same derivatives for every candidate k_exact evaluates (1 KB per entry and hidden
        // layer; 400 MB at 4096 rollouts -- HBM capacity is not a constraint here).  Allocated lazily at the first screened step.
    }
    ctx->screen = ScreenDev{};
    // the input tables keep zeros in the slots the other operand owns; the slot assignment depends on the network's d
    CK(hipMemsetAsync(ctx->d_FpH, 0, (size_t)ctx->cfg.max_obs * 32 * 2, ctx->stream));
    CK(hipMemsetAsync(ctx->d_FqH, 0, (size_t)ctx->cfg.n_traj * 32 * 2, ctx->stream));
    ctx->screen_ok = false;
    ctx->screen_cal = false;
    if (!ctx->screen_eps_fixed) ctx->screen_eps = 0.f;
    ctx->screen_suspended = false;
    ctx->screen_consec = 0;
    ctx->scr_W.clear(); ctx->scr_b.clear(); ctx->scr_out_dims.clear();
    ctx->scr_reorder_pending = false;
    if (!pk.wh.empty()) {
        const uint16_t* dwh = nullptr;
        if ((rc = upload(ctx, pk.wh, &dwh))) return rc;
        if ((rc = upload(ctx, pk.sbias, &ctx->screen.bias))) return rc;
        ctx->screen.Wh = dwh;
        ctx->screen_ok = true;
        ctx->scr_W = std::move(pk.host_W);
        ctx->scr_b = std::move(pk.host_b);
        ctx->scr_out_dims = std::move(pk.out_dims);
        if (skip_mask) {   // the concatenation operands of the screening kernel (omds_screen_sidx), beside FqH / FpH
            const size_t bq = (size_t)ctx->cfg.n_traj * 32 * 2, bp = (size_t)ctx->cfg.max_obs * 32 * 2;
            if (!ctx->d_FqS) CK(hipMalloc(&ctx->d_FqS, bq));
            if (!ctx->d_FpS) CK(hipMalloc(&ctx->d_FpS, bp));
            CK(hipMemsetAsync(ctx->d_FqS, 0, bq, ctx->stream));
            CK(hipMemsetAsync(ctx->d_FpS, 0, bp, ctx->stream));
            m.scrQ = ctx->d_FqS;
            m.scrP = ctx->d_FpS;
        }
    }
    if ((rc = upload(ctx, pk.wf16, &m.Wf16))) return rc;
    if ((rc = upload(ctx, pk.wb16, &m.Wb16))) return rc;
    if ((rc = upload(ctx, pk.wb4, &m.Wb4))) return rc;
    if ((rc = upload(ctx, pk.wf4, &m.Wf4))) return rc;
    if ((rc = upload(ctx, pk.w1b16, &m.W1b16))) return rc;
    if ((rc = upload(ctx, pk.wf, &m.Wf))) return rc;
    if ((rc = upload(ctx, pk.wb, &m.Wb))) return rc;
    if ((rc = upload(ctx, pk.bh, &m.bh))) return rc;
    if ((rc = upload(ctx, pk.wl, &m.Wl))) return rc;
    if ((rc = upload(ctx, pk.bl, &m.bl))) return rc;
    if ((rc = upload(ctx, pk.wlraw, &m.Wlraw))) return rc;
    if ((rc = upload(ctx, pk.whraw, &m.Whraw))) return rc;
    if ((rc = upload(ctx, pk.w1t, &m.W1t))) return rc;
    if ((rc = upload(ctx, pk.b1, &m.b1))) return rc;
    if ((rc = upload(ctx, pk.w1b, &m.W1b))) return rc;
    if ((rc = upload(ctx, pk.w1f, &m.W1f))) return rc;
    if ((rc = upload(ctx, pk.w1f16, &m.W1f16))) return rc;
    if ((rc = upload(ctx, pk.wht, &m.WhT))) return rc;
    if ((rc = upload(ctx, pk.wlt, &m.WlT))) return rc;
    {
        std::vector<unsigned long long> zero(2 * (OMDS_MAX_HIDDEN + 1) + 2, 0ull);
        const unsigned long long* dz = nullptr;
        if ((rc = upload(ctx, zero, &dz))) return rc;
        m.skip_stats = const_cast<unsigned long long*>(dz);
    }
    // the exact zero-skip of k_pass1 (per-tile compaction): every ReLU network without skip concatenations
    m.compact = (act == OMDS_ACT_RELU && skip_mask == 0 && pk.nhh >= 1 && !(ctx->cfg.flags & OMDS_FLAG_DENSE_PASS1)) ? 1 : 0;
#ifdef OMDS_TIMELINE
    {
        static unsigned long long* tl = nullptr;
        if (!tl) { CK(hipMalloc(&tl, (size_t)(1 << 16) * 16 * sizeof(unsigned long long))); }
        CK(hipMemset(tl, 0, (size_t)(1 << 16) * 16 * sizeof(unsigned long long)));
        m.tl = tl;
    }
#endif
    ctx->mlp = m;
    ctx->act = act;
    ctx->f_fwd = pk.f_fwd;
    ctx->f_bwd = pk.f_bwd;
    ctx->have_mlp = true;
    if (ctx->n_obs > 0) {  // re-derive the obstacle half of layer 1 for the new weights
        omds_launch_obstacle_features(ctx->stream, ctx->mlp, ctx->d_obs, ctx->n_obs, ctx->d_Fp, ctx->d_radius, ctx->d_FpH, ctx->cfg.max_obs);
        CK(hipGetLastError());
        CK(hipStreamSynchronize(ctx->stream));
    }
    return OMDS_OK;
}

#ifdef OMDS_TEST_HOOKS
// Test hook (include/omds_test.h): the host half of omds_set_mlp_ex alone -- validation, padding, every fragment pack -- with no
// device and no context, so that the sanitizer build can run it on the CPU.  *checksum = FNV-1a over all packs in upload order,
// *bytes = their total size; message of a failure through omds_last_error(NULL).
int omds_test_pack_mlp(int n_dof, int n_linear, const int32_t* in_dims, const int32_t* out_dims, const float* const* W,
                       const float* const* b, int act, float out_div, int n_skips, const int32_t* skip_after, uint64_t* checksum,
                       int64_t* bytes) {
    MlpPacks pk;
    const int rc = build_mlp_packs(n_dof, n_linear, in_dims, out_dims, W, b, act, out_div, n_skips, skip_after, pk, g_create_err);
    if (rc) return rc;
    uint64_t h = 1469598103934665603ull;
    int64_t total = 0;
    auto mix = [&](const void* p, size_t nb) {
        const unsigned char* c = static_cast<const unsigned char*>(p);
        for (size_t i = 0; i < nb; ++i) { h ^= c[i]; h *= 1099511628211ull; }
        total += (int64_t)nb;
    };
    mix(pk.wh.data(), pk.wh.size() * 2); mix(pk.sbias.data(), pk.sbias.size() * 4);
    mix(pk.wf16.data(), pk.wf16.size() * 16); mix(pk.wb16.data(), pk.wb16.size() * 16); mix(pk.wb4.data(), pk.wb4.size() * 16); mix(pk.wf4.data(), pk.wf4.size() * 16);
    mix(pk.w1b16.data(), pk.w1b16.size() * 16); mix(pk.wf.data(), pk.wf.size() * 16); mix(pk.wb.data(), pk.wb.size() * 16);
    mix(pk.bh.data(), pk.bh.size() * 4); mix(pk.wl.data(), pk.wl.size() * 16); mix(pk.bl.data(), pk.bl.size() * 4);
    mix(pk.wlraw.data(), pk.wlraw.size() * 4); mix(pk.whraw.data(), pk.whraw.size() * 4); mix(pk.w1t.data(), pk.w1t.size() * 4);
    mix(pk.b1.data(), pk.b1.size() * 4); mix(pk.w1b.data(), pk.w1b.size() * 16);
    // the screening pack once more in another unit order (what screen_reorder does behind a calibration): every hidden level reversed.
    // A permuted pack holds the same multiset of weights per slice row set; its bytes go into the same checksum
    if (!pk.host_W.empty() && pk.skip_mask == 0) {
        std::vector<int32_t> order((size_t)(pk.nhh + 1) * OMDS_WIDTH);
        for (int L = 0; L <= pk.nhh; ++L)
            for (int q = 0; q < OMDS_WIDTH; ++q) order[(size_t)L * OMDS_WIDTH + q] = OMDS_WIDTH - 1 - q;
        build_screen_pack(pk, order.data());
        mix(pk.wh.data(), pk.wh.size() * 2); mix(pk.sbias.data(), pk.sbias.size() * 4);
    }
    if (checksum) *checksum = h;
    if (bytes) *bytes = total;
    return OMDS_OK;
}
#endif

// The obstacle buffers grow on demand (MPPI.update_obstacles takes any obstacle count at any time, MPPI.py:347-350): everything
// sized by max_obs is re-allocated at twice the new count; the handle, the network, the policy samples, the communicator and
// the screening state survive.  Nothing of their old contents is needed: the caller is about to replace the scene.
static int grow_obstacle_capacity(omds_ctx* ctx, int n_obs) {
    const size_t N = ctx->cfg.n_traj, k = ctx->cfg.n_closest, H = ctx->cfg.horizon, n = ctx->cfg.n_dof, Km = ctx->cfg.n_kernel_max;
    // twice the new count, but never past what 32-bit pair indices allow: a count that fits is not rejected for its doubling
    const long long Omax = ((1LL << 31) - 1) / (long long)N;
    REQUIRE(n_obs <= Omax, OMDS_ERR_INVALID_ARG, "omds_set_obstacles: n_traj * n_obs must stay below 2^31");
    const size_t Om = (size_t)std::min<long long>(std::max(2LL * n_obs, 64LL), Omax), rows2 = N * k;
    CK(hipStreamSynchronize(ctx->stream));
    // Until every replacement exists the context holds NO scene: if an allocation below fails, check_ready refuses to run
    // (n_obs == 0) and the next omds_set_obstacles starts the growth again (max_obs == 0) instead of touching freed buffers
    ctx->n_obs = 0;
    ctx->cfg.max_obs = 0;
    void** olds[] = {(void**)&ctx->d_obs, (void**)&ctx->d_Fp, (void**)&ctx->d_radius, (void**)&ctx->d_FpH, (void**)&ctx->d_Dmin,
                     (void**)&ctx->d_rowlist, (void**)&ctx->d_listDa, (void**)&ctx->d_FpS};
    const bool had_FpS = ctx->d_FpS != nullptr;
    for (void** o : olds) { if (*o) (void)hipFree(*o); *o = nullptr; }
    CK(hipMalloc(&ctx->d_obs, Om * 4 * 4));
    CK(hipMalloc(&ctx->d_Fp, std::max(Om, rows2) * OMDS_FROW * 4));
    CK(hipMemsetAsync(ctx->d_Fp, 0, std::max(Om, rows2) * OMDS_FROW * 4, ctx->stream));   // the joints' slots and the padding stay zero
    CK(hipMalloc(&ctx->d_radius, std::max(Om, rows2) * 4));
    CK(hipMalloc(&ctx->d_FpH, Om * 32 * 2));
    CK(hipMemsetAsync(ctx->d_FpH, 0, Om * 32 * 2, ctx->stream));
    CK(hipMalloc(&ctx->d_Dmin, N * Om * 4));
    CK(hipMalloc(&ctx->d_rowlist, N * Om * 4));
    CK(hipMalloc(&ctx->d_listDa, N * Om * 4));
    if (had_FpS) {
        CK(hipMalloc(&ctx->d_FpS, Om * 32 * 2));
        CK(hipMemsetAsync(ctx->d_FpS, 0, Om * 32 * 2, ctx->stream));
        ctx->mlp.scrP = ctx->d_FpS;
    }
    const size_t stage = std::max({N * H * std::max(Km, n) * 4, N * Om * 4, Km * n * N * 4, rows2 * OMDS_CPAD * 4});
    if (stage > ctx->stage_bytes) {
        (void)hipFree(ctx->d_stage);
        ctx->d_stage = nullptr;
        ctx->stage_bytes = 0;
        CK(hipMalloc(&ctx->d_stage, stage));
        ctx->stage_bytes = stage;
    }
    const int ex_cap = (int)std::min<size_t>(N * Om, N * 32);
    if (ex_cap > ctx->ex_cap) {
        for (void** o : {(void**)&ctx->d_exD, (void**)&ctx->d_exDr, (void**)&ctx->d_exMin, (void**)&ctx->d_exMask, (void**)&ctx->d_exDeriv}) { if (*o) (void)hipFree(*o); *o = nullptr; }   // (d_exDeriv: allocated again at the next screened tanh step)
        ctx->ex_cap = 0;
        CK(hipMalloc(&ctx->d_exD, (size_t)ex_cap * 4));
        CK(hipMalloc(&ctx->d_exDr, (size_t)ex_cap * 4));
        CK(hipMalloc(&ctx->d_exMin, (size_t)ex_cap * 4));
        CK(hipMalloc(&ctx->d_exMask, (size_t)ex_cap * (OMDS_MAX_HIDDEN + 1) * 8 * 4));
        ctx->ex_cap = ex_cap;
    }
    ctx->cfg.max_obs = (int)Om;
    ctx->n_obs = 0;
    return OMDS_OK;
}

// Has the scene changed enough since the screening bound was calibrated that the calibration batch no longer stands for it?
// Another obstacle count, another radius, or any sphere more than 0.1 (scene units: metres for the Franka scenes) away from
// where it was: a translating / vibrating scene (obstacleStreamer.py:125-137) keeps its bound, a swapped scene does not.
// (Every step of every propagate additionally audits a sample of the unevaluated pairs, omds.h.)
static bool scene_differs_from_calibration(const omds_ctx* ctx, const float* xyzr, int n_obs) {
    if (ctx->obs_cal.size() != (size_t)n_obs * 4) return true;
    for (int i = 0; i < n_obs; ++i) {
        const float* a = &ctx->obs_cal[(size_t)i * 4];
        const float* b = xyzr + (size_t)i * 4;
        for (int c = 0; c < 3; ++c)
            if (!(std::fabs(a[c] - b[c]) <= 0.1f)) return true;
        if (!(std::fabs(a[3] - b[3]) <= 1e-6f)) return true;
    }
    return false;
}

int omds_set_obstacles(omds_ctx* ctx, const float* xyzr, int n_obs) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(xyzr && n_obs >= 1, OMDS_ERR_INVALID_ARG, "omds_set_obstacles: need n_obs >= 1 and a non-null [O,4] array");
    REQUIRE(n_obs >= ctx->cfg.n_closest, OMDS_ERR_INVALID_ARG, "omds_set_obstacles: fewer obstacles than n_closest");
    CK(hipSetDevice(ctx->dev));
    int rc;
    if (n_obs > ctx->cfg.max_obs && (rc = grow_obstacle_capacity(ctx, n_obs))) return rc;
    CK(hipMemcpyAsync(ctx->d_obs, xyzr, (size_t)n_obs * 16, hipMemcpyHostToDevice, ctx->stream));
    ctx->n_obs = n_obs;
    if (ctx->screen_cal && !ctx->screen_eps_fixed && scene_differs_from_calibration(ctx, xyzr, n_obs)) {
        ctx->screen_cal = false;        // calibrate again at the next screened propagate, against THIS scene
        ctx->screen_suspended = false;
        ctx->screen_consec = 0;
    }
    ctx->obs_now.assign(xyzr, xyzr + (size_t)n_obs * 4);
    if (ctx->have_mlp && !ctx->wide.on) {
        omds_launch_obstacle_features(ctx->stream, ctx->mlp, ctx->d_obs, n_obs, ctx->d_Fp, ctx->d_radius, ctx->d_FpH, ctx->cfg.max_obs);
        CK(hipGetLastError());
    }
    CK(hipStreamSynchronize(ctx->stream));
    return OMDS_OK;
}

static void refresh_goal_fk(omds_ctx* ctx) {
    if (ctx->have_ds && ctx->have_cost) omds_host_link_endpoints(ctx->qf, ctx->dh, ctx->cfg.n_dof, ctx->goal_fk);
}

int omds_set_ds(omds_ctx* ctx, const float* q_goal) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(q_goal, OMDS_ERR_INVALID_ARG, "omds_set_ds: null q_goal");
    std::memcpy(ctx->qf, q_goal, ctx->cfg.n_dof * sizeof(float));
    ctx->have_ds = true;
    ctx->have_A = false;
    ctx->seds_G = 0;
    refresh_goal_fk(ctx);
    return OMDS_OK;
}

int omds_set_ds_matrix(omds_ctx* ctx, const float* q_goal, const float* A) {
    int rc = omds_set_ds(ctx, q_goal);
    if (rc || !A) return rc;
    const int n = ctx->cfg.n_dof;
    CK(hipSetDevice(ctx->dev));
    CK(hipMemcpy(ctx->d_A, A, (size_t)n * n * sizeof(float), hipMemcpyHostToDevice));
    ctx->have_A = true;
    return OMDS_OK;
}

int omds_set_ds_seds(omds_ctx* ctx, const float* q_goal, int G, const float* mu_in, const float* b, const float* sigma_inv,
                     const float* A, const float* prior, const float* den, float lin_thr, float seds_thr) {
    int rc = omds_set_ds(ctx, q_goal);
    if (rc || G == 0) return rc;
    REQUIRE(G >= 1 && G <= 64 && mu_in && b && sigma_inv && A && prior && den, OMDS_ERR_INVALID_ARG,
            "omds_set_ds_seds: need 1 <= n_gauss <= 64 and non-null component arrays");
    const int n = ctx->cfg.n_dof, st = omds_seds_stride(n);
    std::vector<float> pk((size_t)G * st);
    for (int j = 0; j < G; ++j) {
        float* p = &pk[(size_t)j * st];
        std::memcpy(p, mu_in + (size_t)j * n, n * sizeof(float));
        std::memcpy(p + n, b + (size_t)j * n, n * sizeof(float));
        p[2 * n] = prior[j];
        p[2 * n + 1] = den[j];
        std::memcpy(p + 2 * n + 2, sigma_inv + (size_t)j * n * n, (size_t)n * n * sizeof(float));
        std::memcpy(p + 2 * n + 2 + n * n, A + (size_t)j * n * n, (size_t)n * n * sizeof(float));
    }
    CK(hipSetDevice(ctx->dev));
    CK(hipStreamSynchronize(ctx->stream));
    if (ctx->d_seds) { (void)hipFree(ctx->d_seds); ctx->d_seds = nullptr; }
    CK(hipMalloc(&ctx->d_seds, pk.size() * sizeof(float)));
    CK(hipMemcpy(ctx->d_seds, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    ctx->seds_G = G;
    ctx->seds_lin_thr = lin_thr;
    ctx->seds_thr = seds_thr;
    return OMDS_OK;
}

int omds_set_params(omds_ctx* ctx, const omds_params* p) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(p, OMDS_ERR_INVALID_ARG, "omds_set_params: null params");
    REQUIRE(p->rbf_p > 0.f, OMDS_ERR_INVALID_ARG, "omds_set_params: rbf_p must be positive");
    REQUIRE((p->cost_terms & ~OMDS_COST_ALL) == 0 && (p->variant & ~3u) == 0, OMDS_ERR_INVALID_ARG,
            "omds_set_params: unknown bits in cost_terms / variant");
    if (p->ignored_links != ctx->prm.ignored_links && !ctx->screen_eps_fixed) {   // another set of links enters the pass-1 minimum
        ctx->screen_cal = false;
        ctx->screen_suspended = false;
        ctx->screen_consec = 0;
    }
    ctx->prm = *p;
    return OMDS_OK;
}

int omds_set_cost(omds_ctx* ctx, const float* dh_params, const float* q_min, const float* q_max) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(dh_params && q_min && q_max, OMDS_ERR_INVALID_ARG, "omds_set_cost: null argument");
    const int n = ctx->cfg.n_dof;
    std::memcpy(ctx->dh, dh_params, (size_t)(n + 1) * 4 * sizeof(float));
    std::memcpy(ctx->qmin, q_min, n * sizeof(float));
    std::memcpy(ctx->qmax, q_max, n * sizeof(float));
    ctx->have_cost = true;
    refresh_goal_fk(ctx);
    return OMDS_OK;
}

// ---- policy samples -------------------------------------------------------------------------------
int omds_set_policy_samples(omds_ctx* ctx, const float* mu, const float* sigma, const float* alpha, int K) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(K >= 0 && K <= ctx->cfg.n_kernel_max, OMDS_ERR_INVALID_ARG, "omds_set_policy_samples: 0 <= n_kernels <= n_kernel_max");
    CK(hipSetDevice(ctx->dev));
    ctx->n_kernels = K;
    if (K == 0) return OMDS_OK;
    REQUIRE(mu && sigma && alpha, OMDS_ERR_INVALID_ARG, "omds_set_policy_samples: null sample array");
    const int N = ctx->cfg.n_traj, n = ctx->cfg.n_dof;
    CK(hipMemcpyAsync(ctx->d_stage, mu, (size_t)N * K * n * 4, hipMemcpyHostToDevice, ctx->stream));
    omds_launch_transpose(ctx->stream, ctx->d_stage, ctx->d_muT, N, K * n);
    CK(hipStreamSynchronize(ctx->stream));
    CK(hipMemcpyAsync(ctx->d_stage, alpha, (size_t)N * K * n * 4, hipMemcpyHostToDevice, ctx->stream));
    omds_launch_transpose(ctx->stream, ctx->d_stage, ctx->d_alphaT, N, K * n);
    CK(hipStreamSynchronize(ctx->stream));
    CK(hipMemcpyAsync(ctx->d_stage, sigma, (size_t)N * K * 4, hipMemcpyHostToDevice, ctx->stream));
    omds_launch_transpose(ctx->stream, ctx->d_stage, ctx->d_sigmaT, N, K);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(ctx->stream));
    return OMDS_OK;
}

int omds_sample_policy(omds_ctx* ctx, const float* mu_c, const float* sigma_c, const float* alpha_c, float mu_s,
                       float sigma_s, float alpha_s, int K, uint64_t seed, int64_t rollout_offset) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(K >= 0 && K <= ctx->cfg.n_kernel_max, OMDS_ERR_INVALID_ARG, "omds_sample_policy: 0 <= n_kernels <= n_kernel_max");
    CK(hipSetDevice(ctx->dev));
    ctx->n_kernels = K;
    if (K == 0) return OMDS_OK;
    REQUIRE(mu_c && sigma_c && alpha_c, OMDS_ERR_INVALID_ARG, "omds_sample_policy: null mean array");
    const int N = ctx->cfg.n_traj, n = ctx->cfg.n_dof;
    // through pinned staging: no stream synchronisation here.  The staging is normally rewritten one planner iteration later,
    // behind the synchronisations of omds_propagate and the update; the event covers back-to-back calls
    CK(hipEventSynchronize(ctx->ev_in_means));
    float* means = ctx->h_in + OMDS_MAX_DOF;
    std::memcpy(means, mu_c, (size_t)K * n * 4);
    std::memcpy(means + (size_t)K * n, sigma_c, (size_t)K * 4);
    std::memcpy(means + (size_t)K * n + K, alpha_c, (size_t)K * n * 4);
    CK(hipMemcpyAsync(ctx->d_means, means, (size_t)K * (2 * n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    CK(hipEventRecord(ctx->ev_in_means, ctx->stream));
    omds_launch_sample(ctx->stream, N, n, K, ctx->d_means, mu_s, sigma_s, alpha_s, seed, rollout_offset, ctx->d_muT,
                       ctx->d_sigmaT, ctx->d_alphaT);
    CK(hipGetLastError());
    return OMDS_OK;
}

int omds_get_policy_samples(omds_ctx* ctx, float* mu, float* sigma, float* alpha) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    CK(hipSetDevice(ctx->dev));
    const int N = ctx->cfg.n_traj, n = ctx->cfg.n_dof, K = ctx->n_kernels;
    if (K == 0) return OMDS_OK;
    if (mu) {
        omds_launch_transpose(ctx->stream, ctx->d_muT, ctx->d_stage, K * n, N);
        CK(hipMemcpyAsync(mu, ctx->d_stage, (size_t)N * K * n * 4, hipMemcpyDeviceToHost, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
    }
    if (alpha) {
        omds_launch_transpose(ctx->stream, ctx->d_alphaT, ctx->d_stage, K * n, N);
        CK(hipMemcpyAsync(alpha, ctx->d_stage, (size_t)N * K * n * 4, hipMemcpyDeviceToHost, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
    }
    if (sigma) {
        omds_launch_transpose(ctx->stream, ctx->d_sigmaT, ctx->d_stage, K, N);
        CK(hipMemcpyAsync(sigma, ctx->d_stage, (size_t)N * K * 4, hipMemcpyDeviceToHost, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
    }
    return OMDS_OK;
}

// ---- distance network on a batch: Fq -> pass 1 -> top-k -> pass 2 ---------------------------------
static int prof_collect(omds_ctx* ctx);
static int prof_begin(omds_ctx* ctx) {
    if (!ctx->prof_on) return OMDS_OK;
    // an event record between two kernels costs ~5.7 us of idle GPU (tools/gap_probe.py: back-to-back launches
    // otherwise start with no gap), so a measurement run brackets only every prof_stride-th launch
    ctx->prof_open = (ctx->prof_seen++ % ctx->prof_stride) == 0;
    if (!ctx->prof_open) return OMDS_OK;
    ProfEvents& p = ctx->prof;
    if (p.used >= 4096) { int rc = prof_collect(ctx); if (rc) return rc; }   // elapsed times are read lazily (omds_prof_read); bound the open events
    if (p.used == p.start.size()) {
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        p.start.push_back(a);
        p.stop.push_back(b);
    }
    CK(hipEventRecord(p.start[p.used], ctx->stream));
    return OMDS_OK;
}
static int prof_end(omds_ctx* ctx, int64_t rows, double flops = -1.0, const char* kernel = "k_pass1") {
    if (!ctx->prof_on || !ctx->prof_open) return OMDS_OK;
    ctx->prof_open = false;
    ProfEvents& p = ctx->prof;
    CK(hipEventRecord(p.stop[p.used], ctx->stream));
    p.used++;
    p.launches++;
    p.rows += rows;
    p.flops += flops >= 0.0 ? flops : (double)rows * ctx->f_fwd;
    p.kernel = kernel;
    return OMDS_OK;
}
static int prof_collect(omds_ctx* ctx) {
    ProfEvents& p = ctx->prof;
    for (size_t i = 0; i < p.used; ++i) {
        float ms = 0.f;
        CK(hipEventSynchronize(p.stop[i]));
        CK(hipEventElapsedTime(&ms, p.start[i], p.stop[i]));
        p.ms += ms;
    }
    p.used = 0;
    return OMDS_OK;
}

static bool small_step_wanted(omds_ctx* ctx);

static int enqueue_network(omds_ctx* ctx, const float* qT, int ldq, int B) {
    const MlpDev& m = ctx->mlp;
    const int O = ctx->n_obs, k = ctx->cfg.n_closest;
    if (ctx->wide.on) {   // a hidden layer wider than 256: the unfused GEMM path (wide_kernels.hip)
        int rcw;
        if ((rcw = prof_begin(ctx))) return rcw;
        if ((rcw = omds_wide_network(ctx, qT, ldq, B))) return rcw;
        if ((rcw = prof_end(ctx, (int64_t)B * O, (double)B * O * ctx->f_fwd + (double)B * k * (ctx->f_fwd + ctx->f_bwd), "k_gemm (wide network)"))) return rcw;
        CK(hipGetLastError());
        return OMDS_OK;
    }
    omds_launch_rollout_features(ctx->stream, m, qT, ldq, B, ctx->d_Fq);
    int rc;
    if (small_step_wanted(ctx)) {   // the arithmetic the step of this context uses: the batch entry point reproduces it bit for bit
        if ((rc = prof_begin(ctx))) return rc;
        omds_launch_net_small(ctx->stream, m, ctx->d_Fp, ctx->d_radius, ctx->d_obs, ctx->d_Fq, O, ctx->prm.ignored_links,
                              ctx->cfg.n_dof, k, qT, ldq, B, ctx->d_gradx, ctx->d_drow, ctx->d_idx, ctx->d_Dmin);
        if ((rc = prof_end(ctx, (int64_t)B * O, (double)B * O * ctx->f_fwd + (double)B * k * ctx->f_bwd, "k_step_small"))) return rc;
        CK(hipGetLastError());
        return OMDS_OK;
    }
    if ((rc = prof_begin(ctx))) return rc;
    omds_launch_pass1(ctx->stream, m, ctx->d_Fq, ctx->d_Fp, ctx->d_radius, O, B, ctx->prm.ignored_links, ctx->d_Dmin);
    if ((rc = prof_end(ctx, (int64_t)B * O))) return rc;
    omds_launch_topk(ctx->stream, ctx->d_Dmin, B, O, k, ctx->d_idx);
    omds_launch_pass2(ctx->stream, m, ctx->d_Fq, ctx->d_Fp, ctx->d_radius, ctx->d_obs, ctx->d_idx, B, k, qT, ldq,
                      ctx->d_gradx, ctx->d_drow, nullptr, nullptr, ctx->d_dscr);
    CK(hipGetLastError());
    return OMDS_OK;
}

static int check_ready(omds_ctx* ctx, bool need_ds) {
    REQUIRE(ctx->have_mlp, OMDS_ERR_NOT_INITIALISED, "distance network not set (omds_set_mlp)");
    REQUIRE(ctx->n_obs > 0, OMDS_ERR_NOT_INITIALISED, "obstacles not set (omds_set_obstacles)");
    if (need_ds) REQUIRE(ctx->have_ds, OMDS_ERR_NOT_INITIALISED, "nominal DS not set (omds_set_ds)");
    return OMDS_OK;
}

// ---- screening (screen_kernel.hip): mode, calibration of eps, the per-step launch sequences ---------------------------------
static bool screen_wanted(omds_ctx* ctx) {
    if (!ctx->screen_ok || ctx->screen_suspended) return false;
    int mode = ctx->screen_mode;
    if (mode < 0) {   // the library's default: the all-fp32 step -- screening is OPT-IN (omds.h); OMDS_SCREEN=0|1|2 sets the default of such contexts
        static int env = -2;
        if (env == -2) { const char* e = getenv("OMDS_SCREEN"); env = e ? atoi(e) : 0; }
        mode = env;
    }
    if (mode <= 0) return false;
    if (mode == 1) return true;
    // 2 = where it pays: once pass 1 is throughput-bound (below that a step is a chain of latency-bound launches and the
    // three extra launches cost more than the fp32 pass)
    return (long long)ctx->cfg.n_traj * ctx->n_obs >= 64LL * 1024 && ctx->n_obs >= 4 * ctx->cfg.n_closest;
}

// The screening pack's unit order from what the network does on the calibration batch.  k_exact (the fp32 tile code, mode 1) is run
// once on a uniform pseudo-random sample of the (state, obstacle) pairs of the B calibration states (their layer-1 halves are in
// d_Fq) and leaves their ReLU masks
// (ExactOut::mask, [entries][hidden levels][8 words]); per hidden level the units are sorted by how many of the sampled 32-pair blocks
// they fired in (ties by index), the pack is built again in that order and copied over the old one.  (The candidates' own masks would be
// there for free, but they are the NEAREST obstacles only: ordered by them, 17 % of the k-chunks of the shipped network are dead for a
// wave; ordered by a uniform sample, 25 %.)  Any order computes the same screening function up to the rounding of the fp32
// accumulation; what the order buys is that k_screen's zero test finds whole 16-unit chunks dead.  The fp32 kernels do not use this
// pack: no returned number changes.
// Called twice per calibration: by the calibration itself on its batch of states (B of them, layer-1 halves in d_Fq) -- so that the
// bound is measured on a sorted pack and the first propagate already runs on one -- and behind the first propagate that is accepted
// afterwards on the states its rollouts ended in (from_rollouts: B = n_traj), which is where the next rollouts will be: the
// calibration batch is deliberately broader than the rollouts, and fewer units are silent on it (3 / 41 / 92 / 120 of the shipped
// network's 256 per layer against 27 / 42 / 93 / 120).
static int screen_reorder(omds_ctx* ctx, int B, bool from_rollouts) {
    static const int enabled = OMDS_EXP_ENV("OMDS_SCREEN_REORDER", 1);   // experiment builds: 0 keeps the natural order, 2 = the calibration's order only (A/B runs)
    const MlpDev& m = ctx->mlp;
    if (from_rollouts) ctx->scr_reorder_pending = false;
    if (!enabled || (enabled == 2 && from_rollouts) || m.act != OMDS_ACT_RELU || m.skip_mask || ctx->scr_W.empty() || !ctx->d_exMask) return OMDS_OK;
    if (from_rollouts)   // layer-1 halves of the last states the propagate reached (d_Fq is rebuilt at the start of every propagate)
        omds_launch_rollout_features(ctx->stream, m, ctx->d_trajT + (size_t)(ctx->cfg.horizon - 1) * ctx->cfg.n_dof * ctx->cfg.n_traj, ctx->cfg.n_traj, B, ctx->d_Fq);
    const int nhid = m.nhh + 1, Wd = OMDS_WIDTH, O = ctx->n_obs;
    const long long pairs = (long long)B * O;
    const int S = (int)std::min<long long>({8192, (long long)ctx->ex_cap, pairs});
    if (S < 64) return OMDS_OK;
    if (!ctx->d_scr_tmp) CK(hipMalloc(&ctx->d_scr_tmp, 8 * sizeof(int)));
    std::vector<int32_t> list(S + 1);
    uint64_t x = 0x9E3779B97F4A7C15ull * (uint64_t)(ctx->screen_recals + 1);
    // the sample is made of BLOCKS of 32 consecutive pairs (one state, 32 consecutive obstacles): what a wave of k_screen multiplies
    // together, and so what a chunk has to be silent for
    for (int j = 0; j < S; j += 32) {   // splitmix64: a fixed sequence per calibration
        x += 0x9E3779B97F4A7C15ull;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        const long long p0 = (long long)(z % (uint64_t)std::max<long long>(pairs - 31, 1));
        for (int i = 0; i < 32 && j + i < S; ++i) list[j + i] = (int32_t)std::min<long long>(p0 + i, pairs - 1);
    }
    list[S] = S;
    // pageable sources: the copies have read them when the calls return
    CK(hipMemcpyAsync(ctx->d_rowlist, list.data(), (size_t)S * 4, hipMemcpyHostToDevice, ctx->stream));
    CK(hipMemsetAsync(ctx->d_scr_tmp, 0, 8 * sizeof(int), ctx->stream));
    CK(hipMemcpyAsync(ctx->d_scr_tmp + 4, &list[S], 4, hipMemcpyHostToDevice, ctx->stream));
    ExactOut ex{ctx->d_exD, ctx->d_exDr, ctx->d_exMin, ctx->d_exMask, ctx->ex_cap};
    omds_launch_exact(ctx->stream, m, ctx->d_Fq, ctx->d_Fp, ctx->d_radius, O, B, ctx->prm.ignored_links, ctx->d_Dmin, ctx->d_rowlist,
                      ctx->d_scr_tmp + 4, reinterpret_cast<unsigned*>(ctx->d_scr_tmp), ex);
    CK(hipGetLastError());
    std::vector<uint32_t> masks((size_t)S * nhid * 8);
    CK(hipMemcpyAsync(masks.data(), ctx->d_exMask, masks.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    std::vector<int32_t> order((size_t)nhid * Wd);
    std::vector<int> count(Wd);
    for (int L = 0; L < nhid; ++L) {
        std::fill(count.begin(), count.end(), 0);
        for (int e0 = 0; e0 < S; e0 += 32) {   // a unit counts once per block it fires in
            uint32_t any[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int e = e0; e < std::min(e0 + 32, S); ++e)
                for (int wi = 0; wi < 8; ++wi) any[wi] |= masks[((size_t)e * nhid + L) * 8 + wi];
            const uint32_t* mr = any;
            for (int wi = 0; wi < 8; ++wi)
                for (uint32_t w = mr[wi]; w; w &= w - 1) {
                    const int bit = __builtin_ctz(w);
                    // level 0: word 2c + h holds the ballot of component c over lanes 32h .. 32h + 31, lane l <-> unit 4l + c;
                    // later levels: word wi holds units 32 wi .. 32 wi + 31 (mlp_device.h)
                    count[L == 0 ? 4 * (32 * (wi & 1) + bit) + (wi >> 1) : 32 * wi + bit]++;
                }
        }
        int32_t* ord = &order[(size_t)L * Wd];
        for (int u = 0; u < Wd; ++u) ord[u] = u;
        std::stable_sort(ord, ord + Wd, [&](int a1, int a2) { return count[a1] > count[a2]; });
        ctx->scr_never_fired[L] = (int)std::count(count.begin(), count.end(), 0);
    }
    MlpPacks pk;
    pk.nhh = m.nhh; pk.C = m.C; pk.d = m.d; pk.act = m.act; pk.skip_mask = m.skip_mask;
    pk.host_W = std::move(ctx->scr_W); pk.host_b = std::move(ctx->scr_b); pk.out_dims = ctx->scr_out_dims;
    build_screen_pack(pk, order.data());
    ctx->scr_W = std::move(pk.host_W); ctx->scr_b = std::move(pk.host_b);
    // the stream is idle (synchronised above, nothing enqueued since): the pack is replaced in place
    CK(hipMemcpy(const_cast<void*>(ctx->screen.Wh), pk.wh.data(), pk.wh.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(const_cast<float*>(ctx->screen.bias), pk.sbias.data(), pk.sbias.size() * 4, hipMemcpyHostToDevice));
    ctx->scr_reorders++;
    return OMDS_OK;
}

// eps = 6 x the largest |screening value - fp32 value| over a calibration batch of up to 1024 states x all obstacles
// (~3e5 pairs: about what one propagate evaluates per step), against the CURRENT obstacle set: half of the states uniform
// inside the joint limits (omds_set_cost) or [-pi, pi], half drawn from the rollouts of the last propagate -- where the next
// rollouts will live -- or, before the first propagate, scattered around the start state (sigma 0.6 rad).  Everything runs on
// the device (k_calib_states -> layer 1 -> k_pass1 and k_screen -> k_max_abs_diff); four bytes come back.  Run at the first
// screened propagate after omds_set_mlp, after omds_set_obstacles with a changed scene (scene_differs_from_calibration), after
// a change of ignored_links and on request (omds_set_screening(mode, eps < 0)).  Between calibrations every propagate
// re-measures the error on its candidates and on the audit sample of the unevaluated pairs (omds_propagate).
static int calibrate_screen(omds_ctx* ctx, const float* q_center) {
    ctx->screen_cal = true;
    ctx->obs_cal = ctx->obs_now;
    const int n = ctx->cfg.n_dof, O = ctx->n_obs;
    const int B = std::min(ctx->cfg.n_traj, 1024);
    float lo[OMDS_MAX_DOF], hi[OMDS_MAX_DOF];
    for (int j = 0; j < n; ++j) {
        lo[j] = ctx->have_cost ? ctx->qmin[j] : -3.14159265f;
        hi[j] = ctx->have_cost ? ctx->qmax[j] : 3.14159265f;
    }
    omds_launch_calib_states(ctx->stream, ctx->d_qstage, B, n, lo, hi, q_center, ctx->have_rollouts ? ctx->d_trajT : nullptr,
                             ctx->cfg.n_traj, ctx->cfg.horizon, 0x9E3779B9u * (unsigned)(ctx->screen_recals + 1));
    omds_launch_rollout_features(ctx->stream, ctx->mlp, ctx->d_qstage, B, B, ctx->d_Fq, ctx->d_FqH, ctx->cfg.n_traj);
    // first the unit order of the screening pack (it follows the scene and the states too), then the bound of THAT pack
    int rc;
    if ((rc = screen_reorder(ctx, B, false))) return rc;
    ctx->scr_reorder_pending = true;
    if (ctx->screen_eps_fixed && ctx->screen_eps > 0.f) return OMDS_OK;   // the bound was set by the caller (omds_set_screening)
    float* apx = ctx->d_stage;   // [B][O] screening values (stage_bytes >= n_traj * max_obs * 4)
    omds_launch_pass1(ctx->stream, ctx->mlp, ctx->d_Fq, ctx->d_Fp, ctx->d_radius, O, B, ctx->prm.ignored_links, ctx->d_Dmin);
    omds_launch_screen(ctx->stream, ctx->screen, ctx->mlp, ctx->d_FqH, ctx->cfg.n_traj, ctx->d_FpH, ctx->cfg.max_obs, ctx->d_radius, O, B, ctx->prm.ignored_links, apx);
    CK(hipMemsetAsync(ctx->d_scerr + 3, 0, 4, ctx->stream));
    omds_launch_max_abs_diff(ctx->stream, ctx->d_Dmin, apx, (long long)B * O, ctx->d_scerr + 3);
    CK(hipGetLastError());
    float worst = 0.f;
    CK(hipMemcpyAsync(&worst, ctx->d_scerr + 3, 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    ctx->screen_recals++;
    ctx->screen_err_seen = 0.f;
    ctx->screen_audit_err_seen = 0.f;
    ctx->screen_sweep_err_seen = 0.f;
    const bool finite = worst < 3.0e38f;
    if (!finite && OMDS_EXP_ENV("OMDS_SCREEN_NOGUARD", 0)) { ctx->screen_eps = 1e-3f; return OMDS_OK; }   // experiment builds: timing of deliberately broken screening kernels
    if (!finite) { ctx->screen_suspended = true; ctx->screen_eps = 0.f; return OMDS_OK; }   // fp16 range exceeded on this scene: the fp32 step until the next calibration
    // The largest error over the ~10^7 pairs of a propagate was seen at up to 2x the calibration batch's maximum (3.8e-3 vs
    // 1.8e-3 .. 2.3e-3 on the shelf scene, depending on the batch drawn): 6x leaves the run-time guard (fallback above
    // eps / 2) room for a 3x larger error, and the accepted propagates then keep eps at >= 4x the largest error they saw --
    // on the shelf both routes end at 1.5e-2.  (8x of an unlucky batch, 1.9e-2, costs 1.2 candidates per rollout and step,
    // which at N = 1024 pushes k_exact past two tiles per CU: 4.8 M against 5.2 M rollout-steps/s.)
    ctx->screen_eps = std::max(6.f * worst, 1e-12f);
    return OMDS_OK;
}

// The one-launch small-scene step (step_small.hip) when the scene qualifies and the batch is small enough that the
// two-kernel step's tail would sit on a fraction of the CUs (at R rollouts per workgroup; beyond ~3 rounds of workgroups
// k_pass1 + the 16/32-row MFMA tail win back what the extra launch costs).
static bool small_step_wanted(omds_ctx* ctx) {
    if (ctx->wide.on || (ctx->cfg.flags & (OMDS_FLAG_UNFUSED_STEP | OMDS_FLAG_TWO_KERNEL_STEP))) return false;
    const int R = omds_step_small_rollouts(ctx->mlp, ctx->cfg.n_dof, ctx->n_obs, ctx->cfg.n_closest);
    if (R <= 0) return false;
    static const int env = OMDS_EXP_ENV("OMDS_SMALL_STEP", -1);   // experiment builds: 0 / 1 overrides the rule below
    if (env == 0) return false;
    if (env > 0) return true;
    return (ctx->cfg.n_traj + R - 1) / R <= 768;
}

// Buffers of the audit sample (allocated at the first screened propagate, grown when the scene or the rate asks for more):
// the list itself -- about N*H*O / one_in entries, room for twice that -- and the layer-1 table of all horizon steps.
static int prepare_audit(omds_ctx* ctx, SelectSink& sk) {
    sk.audit_rows = nullptr; sk.audit_da = nullptr; sk.audit_total = ctx->d_sctotal + (ctx->cfg.horizon + 1); sk.audit_cap = 0;
    sk.audit_mask = 0xffffffffu;
    const long long N = ctx->cfg.n_traj, H = ctx->cfg.horizon, O = ctx->n_obs;
    if (ctx->audit_one_in <= 0 || N * H * O >= (1LL << 31)) return OMDS_OK;   // no audit (or a row space beyond 32-bit indices)
    const long long want = N * H * (2 * O / ctx->audit_one_in + 4);
    if (want > ctx->audit_cap) {
        CK(hipStreamSynchronize(ctx->stream));
        if (ctx->d_audit_rows) (void)hipFree(ctx->d_audit_rows);
        if (ctx->d_audit_da) (void)hipFree(ctx->d_audit_da);
        ctx->d_audit_rows = nullptr; ctx->d_audit_da = nullptr; ctx->audit_cap = 0;
        CK(hipMalloc(&ctx->d_audit_rows, (size_t)want * 4));
        CK(hipMalloc(&ctx->d_audit_da, (size_t)want * 4));
        ctx->audit_cap = (int)std::min<long long>(want, 0x7fffffff);
    }
    if (!ctx->d_FqAll) {
        CK(hipMalloc(&ctx->d_FqAll, (size_t)N * H * OMDS_FROW * 4));
        CK(hipMemsetAsync(ctx->d_FqAll, 0, (size_t)N * H * OMDS_FROW * 4, ctx->stream));   // the obstacles' slots and the padding stay zero
    }
    sk.audit_rows = ctx->d_audit_rows;
    sk.audit_da = ctx->d_audit_da;
    sk.audit_cap = ctx->audit_cap;
    sk.audit_mask = (unsigned)ctx->audit_one_in - 1u;
    return OMDS_OK;
}

// Buffers of a sweep (allocated at the first one): the fp32 values and the screening values of all pairs of ONE step, and the
// statistics every sweep adds to.
static int prepare_sweep(omds_ctx* ctx) {
    const size_t pairs = (size_t)ctx->cfg.n_traj * ctx->cfg.max_obs;
    if (pairs > ctx->sweep_cap) {
        CK(hipStreamSynchronize(ctx->stream));
        if (ctx->d_sweepD) (void)hipFree(ctx->d_sweepD);
        if (ctx->d_sweepDa) (void)hipFree(ctx->d_sweepDa);
        ctx->d_sweepD = nullptr; ctx->d_sweepDa = nullptr; ctx->sweep_cap = 0;
        CK(hipMalloc(&ctx->d_sweepD, pairs * 4));
        CK(hipMalloc(&ctx->d_sweepDa, pairs * 4));
        ctx->sweep_cap = pairs;
    }
    if (!ctx->d_sweep_hist) {
        CK(hipMalloc(&ctx->d_sweep_hist, OMDS_SWEEP_HIST_WORDS * 8));
        CK(hipMemsetAsync(ctx->d_sweep_hist, 0, OMDS_SWEEP_HIST_WORDS * 8, ctx->stream));
    }
    return OMDS_OK;
}

// One step swept: ALL N x O pairs in fp32 (k_pass1 on the step's layer-1 halves) beside all N x O screening values (k_screen in
// matrix mode on the step's fp16 inputs), compared against the tau the step's selection used (d_range): max |Da - D| ->
// d_scerr[3], the distribution of Da - D over the non-candidates -> d_sweep_hist.  Must be enqueued between the step's selection
// and its tail (the tail overwrites the fp16 inputs with the next step's states).
static void enqueue_sweep_of_step(omds_ctx* ctx, const float* fq_step, int N) {
    omds_launch_pass1(ctx->stream, ctx->mlp, fq_step, ctx->d_Fp, ctx->d_radius, ctx->n_obs, N, ctx->prm.ignored_links, ctx->d_sweepD);
    omds_launch_screen(ctx->stream, ctx->screen, ctx->mlp, ctx->d_FqH, ctx->cfg.n_traj, ctx->d_FpH, ctx->cfg.max_obs, ctx->d_radius, ctx->n_obs, N,
                       ctx->prm.ignored_links, ctx->d_sweepDa);
    omds_launch_sweep_hist(ctx->stream, ctx->d_sweepD, ctx->d_sweepDa, ctx->d_range, N, ctx->n_obs, ctx->screen_eps, ctx->d_sweep_hist, ctx->d_scerr + 3);
    ctx->sweep_steps_now++;
}

static int enqueue_rollouts(omds_ctx* ctx, StepArgs& a, bool tail, bool screen) {
    const int N = a.N, H = a.H, n = a.n;
    int rc;
    if (tail && !screen && small_step_wanted(ctx)) {
        omds_launch_rollout_features(ctx->stream, ctx->mlp, ctx->d_trajT, N, N, ctx->d_Fq, nullptr, N);
        for (int i = 1; i <= H; ++i) {
            RoctxRange r1("TAG: evaluate NN_2-5 + Modulation-propagation (fused small-scene step)");
            a.step = i;
            if ((rc = prof_begin(ctx))) return rc;
            omds_launch_step_small(ctx->stream, ctx->mlp, ctx->d_Fp, ctx->d_radius, ctx->d_obs, ctx->d_Fq, ctx->n_obs,
                                   ctx->prm.ignored_links, a);
            if ((rc = prof_end(ctx, (int64_t)N * ctx->n_obs, (double)N * ctx->n_obs * ctx->f_fwd + (double)N * a.k * ctx->f_bwd, "k_step_small"))) return rc;
        }
        return OMDS_OK;
    }
    ExactOut ex{ctx->d_exD, ctx->d_exDr, ctx->d_exMin, ctx->d_exMask, ctx->ex_cap};
    if (tail) {
        // two launches per step: k_pass1 over all (rollout, obstacle) pairs, then the rollout-local tail; with screening
        // k_pass1 becomes k_screen (fp16) + k_select + k_exact (fp32 on the candidates only).
        // (Measured and rejected: independent rollout groups on separate HIP streams for small batches --
        // planar7_1024x32 ran 9.0 M rollout-steps/s on one stream, 7.1 / 3.3 / 2.4 M on 2 / 4 / 8 -- and a two-half
        // ping-pong for large batches with event-chained pass-1 launches so that one half's tail runs under the
        // other half's pass 1: parity-green, but the half-size launches drain twice per step, 1.00 M vs 1.03 M.)
        // With an audit sample the rollout halves of layer 1 of ALL horizon steps are kept ([H][N][256]: step i reads slab
        // i - 1, its tail writes slab i) so that k_audit can re-evaluate pairs of any step at the end; otherwise one slab
        // is updated in place
        float* fq0 = ctx->d_Fq;
        size_t fq_slab = 0;
        SelectSink sink{};
        // ReLU networks: k_exact leaves masks and k_tail_sel runs the backward only; when a rollout's obstacles fit a
        // workgroup's LDS, k_screen selects in its flush phase (no matrix, no k_select)
        const bool relu = ctx->mlp.act == OMDS_ACT_RELU;
        // tanh networks: k_exact hands 1 - h^2 of every candidate's hidden units to k_tail_sel (ExactOut::deriv), so the step has
        // the ReLU step's shape -- no matrix, no k_select, no second forward in the tail.  The buffer is allocated at the first
        // screened tanh step; if that fails (an enormous batch) the step keeps the matrix route (k_exact mode 3 + k_tail)
        if (screen && !relu && !ctx->d_exDeriv) {
            static const int handover = OMDS_EXP_ENV("OMDS_TANH_HANDOVER", 1);   // experiment builds: 0 keeps the matrix route (A/B runs)
            const size_t bytes = (size_t)(ctx->mlp.nhh + 1) * (size_t)ctx->ex_cap * OMDS_WIDTH * 4;
            if (handover && hipMalloc(&ctx->d_exDeriv, bytes) != hipSuccess) { ctx->d_exDeriv = nullptr; (void)hipGetLastError(); }
        }
        const bool list_tail = relu || ctx->d_exDeriv != nullptr;   // k_tail_sel works from k_exact's per-entry outputs
        ex.deriv = relu ? nullptr : ctx->d_exDeriv;
        static const int fuse_env = OMDS_EXP_ENV("OMDS_SCREEN_FUSE_SELECT", 1);   // experiment builds: 0 keeps k_select as its own launch (A/B runs)
        const bool fuse_select = screen && list_tail && fuse_env != 0 && omds_screen_can_select(ctx->n_obs);
        if (screen) {
            CK(hipMemsetAsync(ctx->d_sctotal, 0, (size_t)(H + 2) * 4, ctx->stream));
            CK(hipMemsetAsync(ctx->d_scerr, 0, 16, ctx->stream));
            sink.rowlist = ctx->d_rowlist;
            sink.listDa = fuse_select ? ctx->d_listDa : nullptr;
            sink.range = ctx->d_range;
            sink.k = a.k;
            sink.delta = OMDS_SCREEN_WINDOW * ctx->screen_eps;
            if ((rc = prepare_audit(ctx, sink))) return rc;
            if (sink.audit_rows) { fq0 = ctx->d_FqAll; fq_slab = (size_t)N * OMDS_FROW; }
        }
        ex.Da = sink.listDa;
        SelectSink* d_sinks = static_cast<SelectSink*>(ctx->d_sinks);
        if (screen) {   // the sinks of all steps in one copy (the pinned staging is free: the previous propagate has been synchronised)
            SelectSink* hs = static_cast<SelectSink*>(ctx->h_sinks);
            for (int i = 1; i <= H; ++i) {
                hs[i - 1] = sink;
                hs[i - 1].total = ctx->d_sctotal + (i - 1);
                hs[i - 1].audit_seed = 0x9E3779B9u * ++ctx->audit_counter;
                hs[i - 1].step_row0 = (i - 1) * N;
            }
            CK(hipMemcpyAsync(d_sinks, hs, (size_t)H * sizeof(SelectSink), hipMemcpyHostToDevice, ctx->stream));
        }
        // Every sweep_every-th screened propagate carries a SWEEP: a complete fp32 check of its last horizon step -- or, in the soak
        // mode, of every step (omds_set_screening_sweep).  The audit sample sees every step thinly, a sweep sees a step whole.
        ctx->sweep_now = false;
        ctx->sweep_steps_now = 0;
        // ... and so does the first propagate on a screening pack whose unit order changed after the bound was measured
        // (screened_verdict: the re-sort on the rollouts' own states): the bound is checked on every pair of a step of the new pack
        // before anything else relies on it
        if (screen && ctx->sweep_every > 0 && ((ctx->screen_propagates++ % ctx->sweep_every) == 0 || ctx->sweep_force_next)) {
            if ((rc = prepare_sweep(ctx))) return rc;
            ctx->sweep_now = true;
            ctx->sweep_force_next = false;
        }
        // SMALL BATCHES of the all-fp32 step of a ReLU network run without a second forward: k_pass1 in its emitting mode leaves, for
        // EVERY pair, what pass 2's forward would compute for it (pass-2 distance, arg-min link, ReLU masks: the two forwards are
        // bit-identical), and k_tail_sel selects from the row of Dmin and runs the backward alone.  The chain of a step loses three
        // dependent GEMMs: integrator tick (N = 1) 0.66 -> 0.56 ms per 10-step propagate, planner defaults (N = 40) 1.02 -> 0.95.
        // Up to 24 576 pairs only: the masks cost the pass-1 epilogue 32 ballots + 64 single-lane LDS stores per wave and layer, which
        // a throughput-bound launch cannot hide (k_pass1 +18 % at 1024 x 294, the step 29.5 -> 33.3 ms: EXPERIMENTS.md B.5).
        bool emit = false;
        ExactOut ex_all{};
        static const int emit_env = OMDS_EXP_ENV("OMDS_PASS1_EMIT", 1);   // experiment builds: 0 keeps k_tail (A/B runs)
        if (!screen && relu && emit_env && !(ctx->cfg.flags & OMDS_FLAG_TAIL_FORWARD) && ctx->mlp.skip_mask == 0 && ctx->mlp.nhh >= 1 &&
            omds_tail_sel_supported(n, a.k) && (long long)N * ctx->n_obs <= 24576) {
            const long long pairs = (long long)N * ctx->cfg.max_obs;
            const int nhid = ctx->mlp.nhh + 1;
            if ((pairs > ctx->all_cap || nhid > ctx->all_nhid) && !(ctx->all_failed_pairs == pairs && ctx->all_failed_nhid == nhid)) {
                for (void** o : {(void**)&ctx->d_allDr, (void**)&ctx->d_allMin, (void**)&ctx->d_allMask}) { if (*o) (void)hipFree(*o); *o = nullptr; }
                ctx->all_cap = 0; ctx->all_nhid = 0;
                const size_t mask_bytes = (size_t)pairs * nhid * 32;
                if (mask_bytes <= ((size_t)8 << 30) && hipMalloc(&ctx->d_allDr, (size_t)pairs * 4) == hipSuccess &&
                    hipMalloc(&ctx->d_allMin, (size_t)pairs * 4) == hipSuccess && hipMalloc(&ctx->d_allMask, mask_bytes) == hipSuccess) {
                    ctx->all_cap = pairs; ctx->all_nhid = nhid;
                } else {   // remembered: the same request is not retried at every propagate (hipMalloc / hipFree synchronise the device)
                    (void)hipGetLastError();
                    for (void** o : {(void**)&ctx->d_allDr, (void**)&ctx->d_allMin, (void**)&ctx->d_allMask}) { if (*o) (void)hipFree(*o); *o = nullptr; }
                    ctx->all_failed_pairs = pairs; ctx->all_failed_nhid = nhid;
                }
            }
            if (ctx->all_cap >= pairs && ctx->all_nhid >= nhid) {
                emit = true;
                ex_all = ExactOut{ctx->d_Dmin, ctx->d_allDr, ctx->d_allMin, ctx->d_allMask, (int)std::min<long long>((long long)N * ctx->n_obs, 0x7fffffffLL)};
            }
        }
        omds_launch_rollout_features(ctx->stream, ctx->mlp, ctx->d_trajT, N, N, fq0, screen ? ctx->d_FqH : nullptr, N);
#ifdef OMDS_EXPERIMENT
        static const int corun = OMDS_EXP_ENV("OMDS_EXACT_CORUN", 0);
        static hipStream_t s2 = nullptr;
        static hipEvent_t ev = nullptr, ev_done = nullptr;
        static ExactOut ex2{};
        static unsigned* err2 = nullptr;
#endif
        for (int i = 1; i <= H; ++i) {
            float* fq_i = fq0 + (size_t)(i - 1) * fq_slab;
            float* fq_next = fq0 + (size_t)std::min(i, H - 1) * fq_slab;
            {
                RoctxRange r1("TAG: evaluate NN_2 (forward pass)");
                if ((rc = prof_begin(ctx))) return rc;
                if (screen) {
                    sink = static_cast<const SelectSink*>(ctx->h_sinks)[i - 1];
#ifdef OMDS_EXPERIMENT
                    // Upper-bound experiment for "k_exact under k_screen" (EXPERIMENTS.md, round 5): a REDUNDANT k_exact over the previous
                    // step's candidate list (scratch outputs: the step's own results are untouched) on a second stream, released when
                    // the previous step is done, so that it runs beside this step's k_screen.  What it adds to the iteration is what
                    // k_screen pays for a co-resident k_exact -- the most a flag-driven overlap could get back is k_exact's own 36 us
                    // minus that.
                    if (corun && i >= 2) {
                        if (!s2) {
                            CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
                            CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                            CK(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
                            ex2.cap = ctx->ex_cap;
                            CK(hipMalloc(&ex2.D, (size_t)ex2.cap * 4)); CK(hipMalloc(&ex2.dr, (size_t)ex2.cap * 4)); CK(hipMalloc(&ex2.amin, (size_t)ex2.cap * 4));
                            CK(hipMalloc(&ex2.mask, (size_t)ex2.cap * (OMDS_MAX_HIDDEN + 1) * 8 * 4));
                            CK(hipMalloc(&err2, 16));
                        }
                        CK(hipEventRecord(ev, ctx->stream));
                        CK(hipStreamWaitEvent(s2, ev, 0));
                        omds_launch_exact(s2, ctx->mlp, fq0 + (size_t)(i - 2) * fq_slab, ctx->d_Fp, ctx->d_radius, ctx->n_obs, N, ctx->prm.ignored_links,
                                          ctx->d_Dmin, ctx->d_rowlist, ctx->d_sctotal + (i - 2), err2, ex2);
                        CK(hipEventRecord(ev_done, s2));
                        if (corun == 2 || i == H)   // 2 = control: the same redundant launch IN the main stream's order (the serial cost of one more k_exact)
                            CK(hipStreamWaitEvent(ctx->stream, ev_done, 0));
                    }
#endif
                    omds_launch_screen(ctx->stream, ctx->screen, ctx->mlp, ctx->d_FqH, ctx->cfg.n_traj, ctx->d_FpH, ctx->cfg.max_obs, ctx->d_radius, ctx->n_obs, N,
                                       ctx->prm.ignored_links, ctx->d_Dmin, fuse_select ? d_sinks + (i - 1) : nullptr);
                    if ((rc = prof_end(ctx, (int64_t)N * ctx->n_obs, -1.0, "k_screen"))) return rc;
                    if (!fuse_select) omds_launch_select(ctx->stream, ctx->d_Dmin, N, ctx->n_obs, sink);
                    omds_launch_exact(ctx->stream, ctx->mlp, fq_i, ctx->d_Fp, ctx->d_radius, ctx->n_obs, N,
                                      ctx->prm.ignored_links, ctx->d_Dmin, ctx->d_rowlist, sink.total, ctx->d_scerr, ex);
                } else if (emit) {   // pass 1 leaves pass 2's forward of every pair (pass1_tile mode 6): the tail runs the backward only
                    omds_launch_pass1_emit(ctx->stream, ctx->mlp, ctx->d_Fq, ctx->d_Fp, ctx->d_radius, ctx->n_obs, N,
                                           ctx->prm.ignored_links, ctx->d_Dmin, ex_all);
                    if ((rc = prof_end(ctx, (int64_t)N * ctx->n_obs))) return rc;
                } else {
                    omds_launch_pass1(ctx->stream, ctx->mlp, ctx->d_Fq, ctx->d_Fp, ctx->d_radius, ctx->n_obs, N,
                                      ctx->prm.ignored_links, ctx->d_Dmin);
                    if ((rc = prof_end(ctx, (int64_t)N * ctx->n_obs))) return rc;
                }
            }
            if (ctx->sweep_now && (ctx->sweep_all_steps || i == H)) {
                RoctxRange r4("screening sweep (all pairs of this step in fp32)");
                enqueue_sweep_of_step(ctx, fq_i, N);
            }
            RoctxRange r2("TAG: evaluate NN_3-5 + Modulation-propagation");
            a.step = i;
            if (screen && list_tail)
                omds_launch_tail_sel(ctx->stream, ctx->mlp, ctx->d_Fp, ctx->d_radius, ctx->d_obs, fq_next, ctx->n_obs, a,
                                     ctx->d_rowlist, ctx->d_range, ex, ctx->d_FqH, N, ctx->screen_eps, ctx->d_scerr + 1);
            else if (screen)
                omds_launch_tail(ctx->stream, ctx->mlp, ctx->d_Fp, ctx->d_radius, ctx->d_obs, ctx->d_Dmin, fq_i,
                                 ctx->d_dscr, ctx->n_obs, a, 0, N, ctx->d_FqH, N, fq_next, ctx->d_range, ctx->screen_eps, ctx->d_scerr + 1);
            else if (emit)   // top-k over the rollout's row of Dmin, masks of the k selected pairs, backward, blend, modulation, Euler step
                omds_launch_tail_sel(ctx->stream, ctx->mlp, ctx->d_Fp, ctx->d_radius, ctx->d_obs, ctx->d_Fq, ctx->n_obs, a,
                                     nullptr, nullptr, ex_all, nullptr, 0, 0.f, nullptr);   // (no window, no slack to count: viol = NULL)
            else
                omds_launch_tail(ctx->stream, ctx->mlp, ctx->d_Fp, ctx->d_radius, ctx->d_obs, ctx->d_Dmin, ctx->d_Fq,
                                 ctx->d_dscr, ctx->n_obs, a, 0, N);
        }
        if (screen) {
            // The audit sample of this propagate in one throughput-shaped launch: k_audit on the recorded pairs against the kept
            // layer-1 slabs of all horizon steps -> d_scerr[2] = max (Da - D); then everything the propagate measured about its
            // screening values comes back through pinned memory.  (Measured and rejected in round 4: k_audit on a second,
            // low-priority stream with the verdict deferred to the next call that publishes results, so that the cost and update
            // kernels run beside it -- 5.92-5.97 ms per iteration against 5.86 ms for this form on the same box,
            // profiles/r04_audit_stream_ab.txt: the cross-stream dependency costs more than the 0.07 ms of kernels it overlaps.)
            if (sink.audit_rows) {
                RoctxRange r3("screening audit sample (fp32 re-evaluation of unevaluated pairs)");
                omds_launch_audit(ctx->stream, ctx->mlp, ctx->d_FqAll, ctx->d_Fp, ctx->d_radius, ctx->n_obs, ctx->prm.ignored_links,
                                  sink.audit_rows, sink.audit_da, sink.audit_total, sink.audit_cap, ctx->d_scerr);
            }
            CK(hipGetLastError());
            CK(hipMemcpyAsync(ctx->h_verdict, ctx->d_scerr, 16, hipMemcpyDeviceToHost, ctx->stream));
            CK(hipMemcpyAsync(ctx->h_verdict + 4, ctx->d_sctotal, (size_t)(H + 2) * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
    } else {
        for (int i = 1; i <= H; ++i) {   // MPPI.py:101: H network evaluations, the last velocity is not integrated
            if ((rc = enqueue_network(ctx, ctx->d_trajT + (size_t)(i - 1) * n * N, N, N))) return rc;
            a.step = i;
            omds_launch_modulate(ctx->stream, a);
        }
    }
    return OMDS_OK;
}

static int screened_verdict(omds_ctx* ctx, StepArgs& a, bool tail);
int omds_propagate(omds_ctx* ctx, const float* q_cur, int per_rollout) {
    RoctxRange range("TAG: general propagation");
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(q_cur, OMDS_ERR_INVALID_ARG, "omds_propagate: null q_cur");
    int rc;
    if ((rc = check_ready(ctx, true))) return rc;
    CK(hipSetDevice(ctx->dev));
    const int N = ctx->cfg.n_traj, H = ctx->cfg.horizon, n = ctx->cfg.n_dof;
    // all_traj[:, 0, :] = q_cur  (MPPI.py:99)
    if (per_rollout) {
        CK(hipMemcpyAsync(ctx->d_stage, q_cur, (size_t)N * n * 4, hipMemcpyHostToDevice, ctx->stream));
        omds_launch_transpose(ctx->stream, ctx->d_stage, ctx->d_trajT, N, n);
        CK(hipStreamSynchronize(ctx->stream));  // q_cur is caller memory
    } else {
        // pinned staging + a device slot of its own (d_qcur): no synchronisation, and the policy means that k_sample may not
        // have consumed yet (d_means) stay untouched
        CK(hipEventSynchronize(ctx->ev_in_q));
        std::memcpy(ctx->h_in, q_cur, (size_t)n * 4);
        CK(hipMemcpyAsync(ctx->d_qcur, ctx->h_in, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        CK(hipEventRecord(ctx->ev_in_q, ctx->stream));
        omds_launch_broadcast_q(ctx->stream, ctx->d_qcur, n, N, ctx->d_trajT);
    }
    StepArgs a{};
    a.N = N; a.H = H; a.n = n; a.K = ctx->n_kernels; a.Kmax = ctx->cfg.n_kernel_max; a.k = ctx->cfg.n_closest; a.d = ctx->mlp.d;
    a.trajT = ctx->d_trajT; a.distT = ctx->d_distT; a.dotT = ctx->d_dotT; a.actT = ctx->d_actT; a.normalT = ctx->d_normalT;
    a.kvalT = ctx->d_kvalT; a.qdotT = ctx->d_qdotT; a.maxact = ctx->d_maxact; a.phisum0 = ctx->d_phisum0;
    a.muT = ctx->d_muT; a.sigmaT = ctx->d_sigmaT; a.alphaT = ctx->d_alphaT; a.gradx = ctx->d_gradx; a.drow = ctx->d_drow;
    std::memcpy(a.qf, ctx->qf, sizeof(a.qf));
    a.A = ctx->have_A ? ctx->d_A : nullptr;
    a.seds = ctx->seds_G > 0 ? ctx->d_seds : nullptr;
    a.seds_G = ctx->seds_G; a.seds_lin_thr = ctx->seds_lin_thr; a.seds_thr = ctx->seds_thr;
    a.prm = ctx->prm;
    static const int fused = OMDS_EXP_ENV("OMDS_FUSED_TAIL", 1);   // experiment builds: 0 selects the five-kernel step (the release library: OMDS_FLAG_UNFUSED_STEP)
    // a SEDS nominal DS takes the step of stand-alone kernels: only k_modulate carries that branch (step_device.h)
    const bool tail = fused && !(ctx->cfg.flags & OMDS_FLAG_UNFUSED_STEP) && omds_tail_supported(n, a.k) && ctx->seds_G == 0 && !ctx->wide.on;
    bool screen = tail && screen_wanted(ctx);
    if (screen && !ctx->screen_cal && (rc = calibrate_screen(ctx, q_cur))) return rc;
    screen = screen && ctx->screen_ok && !ctx->screen_suspended && ctx->screen_eps > 0.f;
    if ((rc = enqueue_rollouts(ctx, a, tail, screen))) return rc;
    CK(hipGetLastError());
    ctx->have_cost_vals = false;
    CK(hipStreamSynchronize(ctx->stream));
    ctx->have_rollouts = true;
    if (screen && (rc = screened_verdict(ctx, a, tail))) return rc;
    return OMDS_OK;
}

}  // extern "C"

// What a screened propagate MEASURED about the screening values it relied on (the pinned verdict words, complete once the stream
// has been synchronised):
//   err   = max |Da - D| over every candidate pair (the rows nearest the decision threshold, all re-evaluated);
//   aerr  = max (Da - D) over the audit sample, a uniform pseudo-random 1 in audit_one_in (another one every step) of
//           the pairs that were NOT re-evaluated -- the population the selection rule's assumption is about --
//           evaluated in fp32 by k_audit at the end of the horizon loop;
//   slack = rollouts whose exact k-th smallest candidate came within eps of tau;
//   serr  = (every sweep_every-th propagate) max |Da - D| over ALL pairs of the swept horizon step(s).
// Accepted only while the errors keep a 2x margin to eps and no slack check failed; otherwise the propagate is redone with
// the fp32 pass 1 -- its results are then the fp32 ones by construction -- before omds_propagate returns.
static int screened_verdict(omds_ctx* ctx, StepArgs& a, bool tail) {
    const int N = ctx->cfg.n_traj, H = ctx->cfg.horizon;
    const float* hv = ctx->h_verdict;
    const float err = hv[0], aerr = hv[2], serr = ctx->sweep_now ? hv[3] : 0.f;
    if (ctx->sweep_now) { ctx->screen_sweeps += ctx->sweep_steps_now; if (serr > ctx->screen_sweep_err_seen || serr != serr) ctx->screen_sweep_err_seen = serr; }
    if (err > ctx->screen_err_seen || err != err) ctx->screen_err_seen = err;
    if (aerr > ctx->screen_audit_err_seen || aerr != aerr) ctx->screen_audit_err_seen = aerr;
    const int32_t* tot = reinterpret_cast<const int32_t*>(hv + 4);
    const uint32_t slack_viol = reinterpret_cast<const uint32_t*>(hv)[1];
    bool overflow = false;   // a step listed more rows than k_exact's per-entry outputs hold: redo in fp32
    for (int i = 0; i < H; ++i) { ctx->screen_rows += tot[i]; overflow = overflow || tot[i] > ctx->ex_cap; }
    ctx->screen_audit_rows += std::min<double>(tot[H + 1], ctx->audit_cap);   // entries k_audit evaluated
    ctx->screen_steps += (double)N * H;
    static const int noguard = OMDS_EXP_ENV("OMDS_SCREEN_NOGUARD", 0);   // experiment builds only: the release library cannot switch the guard off
    const float worst = (err != err || aerr != aerr || serr != serr) ? __builtin_inff() : std::max({err, aerr, serr});
    if ((!overflow && worst <= 0.5f * ctx->screen_eps && slack_viol == 0) || noguard) {
        // accepted.  Keep the bound at >= 4x the largest error seen, so that states drifting into regions where the fp16
        // network is less accurate widen it gradually instead of tripping the fallback
        if (!noguard && 4.f * worst > ctx->screen_eps) ctx->screen_eps = 4.f * worst;
        ctx->screen_consec = 0;
        if (ctx->scr_reorder_pending) {   // the unit order once more, on the states the rollouts reached.  eps was measured on the
            // calibration's order: the next propagate carries a sweep (all N x O pairs of a step of the NEW pack in fp32).  The results
            // of this propagate are published already; d_Fq / d_Dmin / d_ex* are scratch between propagates (omds_internal.h)
            const long long before = ctx->scr_reorders;
            const int rrc = screen_reorder(ctx, N, true);
            if (ctx->scr_reorders != before) ctx->sweep_force_next = true;
            return rrc;
        }
        return OMDS_OK;
    }
    // the bound lost its margin on live data (or the list outgrew its buffers): this propagate is redone in fp32 and the bound
    // is widened.  Three in a row: the screening network is not usable on this scene (out of its fp16 range, a scene far from
    // the calibration batch, corrupted weights); the context stays on the fp32 step until the next calibration
    // (omds_set_obstacles with a changed scene, omds_set_mlp, omds_set_screening(mode, eps < 0))
    ctx->screen_fallbacks++;
    if (overflow) ctx->screen_fb_overflow++;
    else if (!(worst <= 0.5f * ctx->screen_eps)) ctx->screen_fb_error++;
    else ctx->screen_fb_slack++;
    if (!overflow && worst < 3.0e38f) ctx->screen_eps = std::max(ctx->screen_eps, 4.f * worst);
    if (++ctx->screen_consec >= 3 || !(worst < 3.0e38f)) { if (!ctx->screen_suspended) ctx->screen_suspensions++; ctx->screen_suspended = true; }
    int rc;
    if ((rc = enqueue_rollouts(ctx, a, tail, false))) return rc;   // from trajT[0], which no step overwrites
    CK(hipGetLastError());
    CK(hipStreamSynchronize(ctx->stream));
    return OMDS_OK;
}

extern "C" {

int omds_get_rollouts(omds_ctx* ctx, float* all_traj, float* closest_dist_all, float* kernel_val_all, float* dot_products,
                      float* kernel_activations, float* qdot, float* normal) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    CK(hipSetDevice(ctx->dev));
    const int N = ctx->cfg.n_traj, H = ctx->cfg.horizon, n = ctx->cfg.n_dof, K = ctx->n_kernels, Km = ctx->cfg.n_kernel_max;
    auto fetch = [&](const float* srcT, float* dst, int X, int Xld) -> int {
        if (!dst || X == 0) return OMDS_OK;
        omds_launch_permute_hxn_to_nhx(ctx->stream, srcT, ctx->d_stage, H, X, N, Xld);
        CK(hipMemcpyAsync(dst, ctx->d_stage, (size_t)N * H * X * 4, hipMemcpyDeviceToHost, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
        return OMDS_OK;
    };
    int rc;
    if ((rc = fetch(ctx->d_trajT, all_traj, n, n))) return rc;
    if ((rc = fetch(ctx->d_distT, closest_dist_all, 1, 1))) return rc;
    if ((rc = fetch(ctx->d_kvalT, kernel_val_all, K, Km))) return rc;
    if ((rc = fetch(ctx->d_dotT, dot_products, 1, 1))) return rc;
    if ((rc = fetch(ctx->d_actT, kernel_activations, 1, 1))) return rc;
    if ((rc = fetch(ctx->d_normalT, normal, n, n))) return rc;
    if (qdot) {
        omds_launch_transpose(ctx->stream, ctx->d_qdotT, ctx->d_stage, n, N);
        CK(hipMemcpyAsync(qdot, ctx->d_stage, (size_t)N * n * 4, hipMemcpyDeviceToHost, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
    }
    return OMDS_OK;
}

// The rows of a few rollouts (reference layouts): what a planner loop reads per iteration -- the best rollout for its FK
// payload, the rollout a new kernel's centre came from (frankaPlanner.py:147-168) -- without moving the N x H tensors.
int omds_get_rollout_rows(omds_ctx* ctx, const int32_t* t, int count, float* all_traj, float* closest_dist_all, float* kernel_val_all,
                          float* dot_products, float* kernel_activations, float* qdot, float* normal) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    const int N = ctx->cfg.n_traj, H = ctx->cfg.horizon, n = ctx->cfg.n_dof, K = ctx->n_kernels, Km = ctx->cfg.n_kernel_max;
    REQUIRE(t && count >= 1 && count <= N, OMDS_ERR_INVALID_ARG, "omds_get_rollout_rows: need 1 <= count <= n_traj and a non-null index array");
    for (int r = 0; r < count; ++r) REQUIRE(t[r] >= 0 && t[r] < N, OMDS_ERR_INVALID_ARG, "omds_get_rollout_rows: rollout index out of range");
    REQUIRE(ctx->have_rollouts, OMDS_ERR_NOT_INITIALISED, "omds_get_rollout_rows: no rollouts yet (omds_propagate)");
    CK(hipSetDevice(ctx->dev));
    const size_t per = (size_t)H * (2 * n + 3 + K) + n;   // floats per rollout over all seven outputs
    REQUIRE((per * count + count) * 4 <= ctx->stage_bytes, OMDS_ERR_INVALID_ARG, "omds_get_rollout_rows: too many rollouts for the staging buffer");
    int* d_t = reinterpret_cast<int*>(ctx->d_stage + per * count);
    CK(hipMemcpyAsync(d_t, t, (size_t)count * 4, hipMemcpyHostToDevice, ctx->stream));
    struct Out { const float* src; float* dst; int Hh, X, Xld; };
    const Out outs[] = {{ctx->d_trajT, all_traj, H, n, n}, {ctx->d_distT, closest_dist_all, H, 1, 1}, {ctx->d_kvalT, kernel_val_all, H, K, Km},
                        {ctx->d_dotT, dot_products, H, 1, 1}, {ctx->d_actT, kernel_activations, H, 1, 1}, {ctx->d_qdotT, qdot, 1, n, n},
                        {ctx->d_normalT, normal, H, n, n}};
    size_t off = 0, offs[7];
    for (int i = 0; i < 7; ++i) {
        offs[i] = off;
        if (!outs[i].dst || outs[i].X == 0) continue;
        omds_launch_gather_rows(ctx->stream, outs[i].src, ctx->d_stage + off, d_t, count, outs[i].Hh, outs[i].X, N, outs[i].Xld);
        off += (size_t)count * outs[i].Hh * outs[i].X;
    }
    CK(hipGetLastError());
    std::vector<float> host(off);
    if (off) CK(hipMemcpyAsync(host.data(), ctx->d_stage, off * 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 7; ++i)
        if (outs[i].dst && outs[i].X) std::memcpy(outs[i].dst, host.data() + offs[i], (size_t)count * outs[i].Hh * outs[i].X * 4);
    return OMDS_OK;
}

int omds_dist_grad(omds_ctx* ctx, const float* q, int B, float* distance, float* nn_grad, float* mindist,
                   int32_t* closest_idx) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(q && B >= 1 && B <= ctx->cfg.n_traj, OMDS_ERR_INVALID_ARG, "omds_dist_grad: need 1 <= batch <= n_traj and non-null q");
    int rc;
    if ((rc = check_ready(ctx, false))) return rc;
    CK(hipSetDevice(ctx->dev));
    const int n = ctx->cfg.n_dof, k = ctx->cfg.n_closest, O = ctx->n_obs, d = ctx->mlp.d;
    CK(hipMemcpyAsync(ctx->d_stage, q, (size_t)B * n * 4, hipMemcpyHostToDevice, ctx->stream));
    omds_launch_transpose(ctx->stream, ctx->d_stage, ctx->d_qstage, B, n);   // -> [n][B]
    if ((rc = enqueue_network(ctx, ctx->d_qstage, B, B))) return rc;
    omds_launch_blend(ctx->stream, ctx->d_gradx, ctx->d_drow, B, k, d, n, ctx->prm.softmax_k, ctx->d_dist, ctx->d_nngrad);
    CK(hipGetLastError());
    if (distance) CK(hipMemcpyAsync(distance, ctx->d_dist, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (nn_grad) CK(hipMemcpyAsync(nn_grad, ctx->d_nngrad, (size_t)B * n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (mindist) CK(hipMemcpyAsync(mindist, ctx->d_Dmin, (size_t)B * O * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (closest_idx) CK(hipMemcpyAsync(closest_idx, ctx->d_idx, (size_t)B * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    return OMDS_OK;
}

// the raw rows x [B][d] through the network and the vjp of one output per row: the arg-min one (seed_col < 0) or column seed_col
static int mlp_rows_vjp(omds_ctx* ctx, const char* who, const float* x, int B, int seed_col, float* y, float* grad, int32_t* min_idx) {
    const int n = ctx->cfg.n_dof;
    const int cap = ctx->cfg.n_traj * ctx->cfg.n_closest;
    if (!(x && B >= 1 && B <= cap)) { ctx->err = std::string(who) + ": need 1 <= batch <= n_traj*n_closest and non-null x"; return OMDS_ERR_INVALID_ARG; }
    REQUIRE(ctx->have_mlp, OMDS_ERR_NOT_INITIALISED, "distance network not set (omds_set_mlp)");
    const int d = ctx->mlp.d;
    CK(hipSetDevice(ctx->dev));
    if (ctx->wide.on) {   // wide networks: the raw rows through the unfused GEMM path
        int rcw;
        CK(hipMemcpyAsync(ctx->d_stage, x, (size_t)B * d * 4, hipMemcpyHostToDevice, ctx->stream));
        if ((rcw = omds_wide_vjp(ctx, ctx->d_stage, B, seed_col))) return rcw;
    } else {
        // every row is its own (rollout, obstacle) pair: Fq from x[:, :n], Fp from x[:, n:], radius 0
        std::vector<float> xyzr((size_t)B * 4, 0.f), qrow((size_t)B * n);
        std::vector<int32_t> ident(B);
        for (int r = 0; r < B; ++r) {
            for (int j = 0; j < n; ++j) qrow[(size_t)r * n + j] = x[(size_t)r * d + j];
            for (int j = 0; j < d - n; ++j) xyzr[(size_t)r * 4 + j] = x[(size_t)r * d + n + j];
            ident[r] = r;
        }
        // per-row "obstacle" buffers of this entry point, allocated on first use for the context's capacity and kept
        if (!ctx->d_vjp_xyzr) {
            CK(hipMalloc(&ctx->d_vjp_xyzr, (size_t)cap * 16));
            CK(hipMalloc(&ctx->d_vjp_B, (size_t)cap * OMDS_FROW * 4));
            CK(hipMemsetAsync(ctx->d_vjp_B, 0, (size_t)cap * OMDS_FROW * 4, ctx->stream));
            ctx->vjp_cap = cap;
            CK(hipMalloc(&ctx->d_vjp_rad, (size_t)cap * 4));
        }
        float *d_xyzr = ctx->d_vjp_xyzr, *d_B = ctx->d_vjp_B, *d_rad = ctx->d_vjp_rad;
        // pageable sources: the copies have read them when the calls return
        CK(hipMemcpyAsync(d_xyzr, xyzr.data(), (size_t)B * 16, hipMemcpyHostToDevice, ctx->stream));
        CK(hipMemcpyAsync(ctx->d_stage, qrow.data(), (size_t)B * n * 4, hipMemcpyHostToDevice, ctx->stream));
        CK(hipMemcpyAsync(ctx->d_idx, ident.data(), (size_t)B * 4, hipMemcpyHostToDevice, ctx->stream));
        omds_launch_transpose(ctx->stream, ctx->d_stage, ctx->d_qstage, B, n);
        omds_launch_rollout_features(ctx->stream, ctx->mlp, ctx->d_qstage, B, B, ctx->d_Fq);
        omds_launch_obstacle_features(ctx->stream, ctx->mlp, d_xyzr, B, d_B, d_rad);
        omds_launch_pass2(ctx->stream, ctx->mlp, ctx->d_Fq, d_B, d_rad, d_xyzr, ctx->d_idx, B, 1, ctx->d_qstage, B,
                          ctx->d_gradx, ctx->d_drow, ctx->d_yraw, ctx->d_minidx, ctx->d_dscr, seed_col);
    }
    CK(hipGetLastError());
    CK(hipStreamSynchronize(ctx->stream));
    if (y) {
        std::vector<float> ypad((size_t)B * OMDS_CPAD);
        CK(hipMemcpy(ypad.data(), ctx->d_yraw, ypad.size() * 4, hipMemcpyDeviceToHost));
        for (int r = 0; r < B; ++r)
            for (int c = 0; c < ctx->mlp.C; ++c) y[(size_t)r * ctx->mlp.C + c] = ypad[(size_t)r * OMDS_CPAD + c];
    }
    if (grad) CK(hipMemcpy(grad, ctx->d_gradx, (size_t)B * d * 4, hipMemcpyDeviceToHost));
    if (min_idx) CK(hipMemcpy(min_idx, ctx->d_minidx, (size_t)B * 4, hipMemcpyDeviceToHost));
    return OMDS_OK;
}

int omds_mlp_forward_vjp(omds_ctx* ctx, const float* x, int B, float* y, float* grad, int32_t* min_idx) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    return mlp_rows_vjp(ctx, "omds_mlp_forward_vjp", x, B, -1, y, grad, min_idx);
}

int omds_mlp_jacobian(omds_ctx* ctx, const float* x, int B, const int32_t* cols, int n_cols, float* y, float* jac) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(ctx->have_mlp, OMDS_ERR_NOT_INITIALISED, "distance network not set (omds_set_mlp)");
    REQUIRE(cols && jac && n_cols >= 1 && n_cols <= OMDS_CPAD, OMDS_ERR_INVALID_ARG, "omds_mlp_jacobian: need cols, jac and 1 <= n_cols <= 16");
    for (int k = 0; k < n_cols; ++k)
        REQUIRE(cols[k] >= 0 && cols[k] < ctx->mlp.C, OMDS_ERR_INVALID_ARG, "omds_mlp_jacobian: a column index is outside 0 .. out_channels - 1");
    const int d = ctx->mlp.d;
    std::vector<float> g;
    for (int k = 0; k < n_cols; ++k) {   // one backward per column, like the reference's loop of .backward() calls (robot_sdf.py:94-100)
        int rc = mlp_rows_vjp(ctx, "omds_mlp_jacobian", x, B, cols[k], k == 0 ? y : nullptr, nullptr, nullptr);
        if (rc) return rc;
        g.resize((size_t)B * d);
        CK(hipMemcpy(g.data(), ctx->d_gradx, g.size() * 4, hipMemcpyDeviceToHost));
        for (int r = 0; r < B; ++r)
            for (int j = 0; j < d; ++j) jac[((size_t)r * d + j) * n_cols + k] = g[(size_t)r * d + j];
    }
    return OMDS_OK;
}

// ---- cost and the cost-weighted update --------------------------------------------------------------
static int enqueue_cost(omds_ctx* ctx) {
    CostArgs a{};
    a.N = ctx->cfg.n_traj; a.H = ctx->cfg.horizon; a.n = ctx->cfg.n_dof;
    a.trajT = ctx->d_trajT; a.distT = ctx->d_distT; a.cost = ctx->d_cost;
    a.terms = ctx->prm.cost_terms;
    std::memcpy(a.qf, ctx->qf, sizeof(a.qf));
    std::memcpy(a.qmin, ctx->qmin, sizeof(a.qmin));
    std::memcpy(a.qmax, ctx->qmax, sizeof(a.qmax));
    std::memcpy(a.dh, ctx->dh, sizeof(a.dh));
    std::memcpy(a.goal_fk, ctx->goal_fk, sizeof(a.goal_fk));
    omds_launch_cost(ctx->stream, a);
    CK(hipGetLastError());
    ctx->have_cost_vals = true;
    return OMDS_OK;
}

int omds_cost(omds_ctx* ctx, float* cost_out) {
    RoctxRange range("TAG: cost calculation");
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(ctx->have_ds && ctx->have_cost, OMDS_ERR_NOT_INITIALISED, "omds_cost: call omds_set_ds and omds_set_cost first");
    CK(hipSetDevice(ctx->dev));
    int rc;
    if ((rc = enqueue_cost(ctx))) return rc;
    if (cost_out) {
        CK(hipMemcpyAsync(cost_out, ctx->d_cost, (size_t)ctx->cfg.n_traj * 4, hipMemcpyDeviceToHost, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
    }
    return OMDS_OK;
}

// Cost.evaluate_costs on caller-supplied tensors (cost.py:13-22 evaluates exactly its arguments): all_traj [B,H,n],
// closest_dist_all [B,H] in the reference layout, B <= n_traj.  The device rollouts of the context are not touched.
int omds_cost_eval(omds_ctx* ctx, const float* all_traj, const float* closest_dist_all, int B, float* cost_out) {
    RoctxRange range("TAG: cost calculation");
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(all_traj && closest_dist_all && cost_out && B >= 1 && B <= ctx->cfg.n_traj, OMDS_ERR_INVALID_ARG,
            "omds_cost_eval: need 1 <= batch <= n_traj and non-null arrays");
    REQUIRE(ctx->have_ds && ctx->have_cost, OMDS_ERR_NOT_INITIALISED, "omds_cost_eval: call omds_set_ds and omds_set_cost first");
    CK(hipSetDevice(ctx->dev));
    const size_t H = ctx->cfg.horizon, n = ctx->cfg.n_dof, N = ctx->cfg.n_traj;
    if (!ctx->d_evalT) CK(hipMalloc(&ctx->d_evalT, (H * n * N + H * N + N) * 4));
    float* trajT = ctx->d_evalT;                 // [H][n][B]
    float* distT = trajT + H * n * N;            // [H][B]
    float* costv = distT + H * N;                // [B]
    CK(hipMemcpyAsync(ctx->d_stage, all_traj, (size_t)B * H * n * 4, hipMemcpyHostToDevice, ctx->stream));
    omds_launch_transpose(ctx->stream, ctx->d_stage, trajT, B, (int)(H * n));
    CK(hipStreamSynchronize(ctx->stream));
    CK(hipMemcpyAsync(ctx->d_stage, closest_dist_all, (size_t)B * H * 4, hipMemcpyHostToDevice, ctx->stream));
    omds_launch_transpose(ctx->stream, ctx->d_stage, distT, B, (int)H);
    CostArgs a{};
    a.N = B; a.H = (int)H; a.n = (int)n;
    a.trajT = trajT; a.distT = distT; a.cost = costv;
    a.terms = ctx->prm.cost_terms;
    std::memcpy(a.qf, ctx->qf, sizeof(a.qf));
    std::memcpy(a.qmin, ctx->qmin, sizeof(a.qmin));
    std::memcpy(a.qmax, ctx->qmax, sizeof(a.qmax));
    std::memcpy(a.dh, ctx->dh, sizeof(a.dh));
    std::memcpy(a.goal_fk, ctx->goal_fk, sizeof(a.goal_fk));
    omds_launch_cost(ctx->stream, a);
    CK(hipGetLastError());
    CK(hipMemcpyAsync(cost_out, costv, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    return OMDS_OK;
}

// Local [sum(cost), N] of this shard.
int omds_cost_sum(omds_ctx* ctx, float* out2) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(out2, OMDS_ERR_INVALID_ARG, "omds_cost_sum: null output");
    REQUIRE(ctx->have_cost_vals, OMDS_ERR_NOT_INITIALISED, "no cost available: call omds_cost after omds_propagate");
    CK(hipSetDevice(ctx->dev));
    const int rs = omds_red_size(ctx->n_kernels, ctx->cfg.n_dof);
    float* red2 = ctx->d_red + rs;  // [sum cost, N] lives behind the packed buffer
    omds_launch_cost_sum(ctx->stream, ctx->d_cost, ctx->cfg.n_traj, red2);
    CK(hipGetLastError());
    CK(hipMemcpyAsync(ctx->h_red + rs, red2, 8, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    out2[0] = ctx->h_red[rs];
    out2[1] = ctx->h_red[rs + 1];
    return OMDS_OK;
}

int omds_red_count(const omds_ctx* ctx) { return ctx ? omds_red_size(ctx->n_kernels, ctx->cfg.n_dof) : 0; }

// Packed partial sums of this shard for the GLOBAL beta = (sum_cost / n_total) / 50.
int omds_local_sums(omds_ctx* ctx, float sum_cost, float n_total, int include_rollout0, float* red_out) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(red_out && n_total > 0.f, OMDS_ERR_INVALID_ARG, "omds_local_sums: null output or n_total <= 0");
    REQUIRE(ctx->have_cost_vals, OMDS_ERR_NOT_INITIALISED, "no cost available: call omds_cost after omds_propagate");
    CK(hipSetDevice(ctx->dev));
    const int N = ctx->cfg.n_traj, n = ctx->cfg.n_dof, K = ctx->n_kernels;
    const int rs = omds_red_size(K, n);
    float* red2 = ctx->d_red + rs;
    ctx->h_red[rs] = sum_cost;
    ctx->h_red[rs + 1] = n_total;
    CK(hipMemcpyAsync(red2, ctx->h_red + rs, 8, hipMemcpyHostToDevice, ctx->stream));
    omds_launch_weights(ctx->stream, ctx->d_cost, N, red2, ctx->d_w, nullptr);
    omds_launch_policy_sums(ctx->stream, N, n, K, ctx->d_w, ctx->d_muT, ctx->d_sigmaT, ctx->d_alphaT, ctx->d_maxact,
                            ctx->d_phisum0, ctx->d_qdotT, ctx->d_cost, include_rollout0, ctx->d_red);
    CK(hipGetLastError());
    CK(hipMemcpyAsync(ctx->h_red, ctx->d_red, (size_t)rs * 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    std::memcpy(red_out, ctx->h_red, (size_t)rs * 4);
    return OMDS_OK;
}

// Pure host arithmetic, no context: masks + theta_c update from the (globally) reduced buffer.
int omds_apply_update(int K, int n, int H, const float* red, float n_total, float rate, float ker_thr, uint32_t variant,
                      float* mu_c, float* sigma_c, float* alpha_c, int32_t* mask_out) {
    if (K < 0 || n < 1 || H < 1 || !red || n_total <= 0.f) return OMDS_ERR_INVALID_ARG;
    if (K > 0 && (!mu_c || !sigma_c || !alpha_c)) return OMDS_ERR_INVALID_ARG;
    const float sumw = red[0];
    const float *s_mu = red + 1, *s_sg = s_mu + K * n, *s_al = s_sg + K, *s_mx = s_al + K * n, *s_ph = s_mx + K;
    for (int kk = 0; kk < K; ++kk) {
        // mask 1: mean over ALL rollouts of max_h(phi*act) > ker_thr; mask 2: mean_h phi of rollout 0 (MPPI.py:336-342)
        const float m1 = s_mx[kk] / n_total, m2 = s_ph[kk] / (float)H;
        const bool upd = (m1 > ker_thr) && ((variant & OMDS_VARIANT_NO_BASE_MASK) || (m2 > ker_thr));   // NaN compares false, like torch
        if (mask_out) mask_out[kk] = upd ? 1 : 0;
        const float u = upd ? rate : 0.f;
        for (int j = 0; j < n; ++j) {
            mu_c[kk * n + j] = (1.f - u) * mu_c[kk * n + j] + u * (s_mu[kk * n + j] / sumw);
            alpha_c[kk * n + j] = (1.f - u) * alpha_c[kk * n + j] + u * (s_al[kk * n + j] / sumw);
        }
        sigma_c[kk] = (1.f - u) * sigma_c[kk] + u * (s_sg[kk] / sumw);
    }
    return OMDS_OK;
}

int omds_weighted_update(omds_ctx* ctx, float rate, float ker_thr, float* mu_c, float* sigma_c, float* alpha_c,
                         int32_t* mask_out, float* weights_out) {
    RoctxRange range("shift_policy_means");
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    int rc;
    if ((rc = omds_update_impl(ctx, false, rate, ker_thr, mu_c, sigma_c, alpha_c, mask_out, nullptr, nullptr, nullptr))) return rc;
    if (weights_out) {
        const int N = ctx->cfg.n_traj;
        std::vector<float> w(N);
        CK(hipMemcpy(w.data(), ctx->d_w, (size_t)N * 4, hipMemcpyDeviceToHost));
        for (int t = 0; t < N; ++t) weights_out[t] = w[t] / ctx->h_red[0];
    }
    return OMDS_OK;
}

// MPPI.shift_policy_means + TensorPolicyMPPI.update_policy (MPPI.py:331-345, policy.py:88-113) on caller-supplied tensors, the way
// omds_cost_eval serves Cost.evaluate_costs: cost [N], kernel_val_all [N,H,K], kernel_activations [N,H] in the reference layouts, against
// the policy samples the context holds.  The context's own rollouts, cost values and running maxima are not touched.
int omds_weighted_update_eval(omds_ctx* ctx, const float* cost, const float* kernel_val_all, const float* kernel_activations, float rate,
                              float ker_thr, float* mu_c, float* sigma_c, float* alpha_c, int32_t* mask_out, float* weights_out) {
    RoctxRange range("shift_policy_means");
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    const int N = ctx->cfg.n_traj, H = ctx->cfg.horizon, n = ctx->cfg.n_dof, K = ctx->n_kernels, Km = ctx->cfg.n_kernel_max;
    REQUIRE(cost && kernel_activations && (K == 0 || (kernel_val_all && mu_c && sigma_c && alpha_c)), OMDS_ERR_INVALID_ARG,
            "omds_weighted_update_eval: null argument");
    CK(hipSetDevice(ctx->dev));
    if (!ctx->d_uev) CK(hipMalloc(&ctx->d_uev, ((size_t)N + (size_t)Km * N + Km + (size_t)N * H) * 4));
    float* d_costv = ctx->d_uev;
    float* d_maxact = d_costv + N;
    float* d_phisum0 = d_maxact + (size_t)Km * N;
    float* d_act = d_phisum0 + Km;
    CK(hipMemcpyAsync(d_costv, cost, (size_t)N * 4, hipMemcpyHostToDevice, ctx->stream));
    CK(hipMemcpyAsync(d_act, kernel_activations, (size_t)N * H * 4, hipMemcpyHostToDevice, ctx->stream));
    if (K > 0) {
        CK(hipMemcpyAsync(ctx->d_stage, kernel_val_all, (size_t)N * H * K * 4, hipMemcpyHostToDevice, ctx->stream));   // stage_bytes >= N*H*Kmax*4
        omds_launch_update_inputs(ctx->stream, ctx->d_stage, d_act, N, H, K, (ctx->prm.variant & OMDS_VARIANT_KVAL_TIMES_ACT) ? 1 : 0, d_maxact, d_phisum0);
    }
    const int rs = omds_red_size(K, n);
    float* red2 = ctx->d_red + rs;
    omds_launch_cost_sum(ctx->stream, d_costv, N, red2);
    omds_launch_weights(ctx->stream, d_costv, N, red2, ctx->d_w, nullptr);
    omds_launch_policy_sums(ctx->stream, N, n, K, ctx->d_w, ctx->d_muT, ctx->d_sigmaT, ctx->d_alphaT, d_maxact, d_phisum0, ctx->d_qdotT, d_costv, 1, ctx->d_red);
    CK(hipGetLastError());
    CK(hipMemcpyAsync(ctx->h_red, ctx->d_red, (size_t)(rs + 2) * 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    const int rc = omds_apply_update(K, n, H, ctx->h_red, ctx->h_red[rs + 1], rate, ker_thr, ctx->prm.variant, mu_c, sigma_c, alpha_c, mask_out);
    if (rc) { ctx->err = "omds_apply_update: invalid argument"; return rc; }
    if (weights_out) {
        std::vector<float> w(N);
        CK(hipMemcpy(w.data(), ctx->d_w, (size_t)N * 4, hipMemcpyDeviceToHost));
        for (int t = 0; t < N; ++t) weights_out[t] = w[t] / ctx->h_red[0];
    }
    return OMDS_OK;
}

int omds_get_qdot(omds_ctx* ctx, int mode, float* out) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(out && (mode == 0 || mode == 1), OMDS_ERR_INVALID_ARG, "omds_get_qdot: mode 0 ('best') or 1 ('weighted'), non-null out");
    // the sums of this shard only (means untouched: rate 0 on null arrays is not allowed, so run the reduction alone)
    int rc;
    float cs[2];
    if ((rc = omds_cost_sum(ctx, cs))) return rc;
    const int n = ctx->cfg.n_dof, K = ctx->n_kernels;
    std::vector<float> red(omds_red_size(K, n));
    if ((rc = omds_local_sums(ctx, cs[0], cs[1], 1, red.data()))) return rc;
    const int o_qd = 1 + K * (2 * n + 3), o_best = o_qd + n;
    for (int j = 0; j < n; ++j) out[j] = mode == 1 ? red[o_qd + j] / red[0] : red[o_best + 1 + j];
    return OMDS_OK;
}

// ---- navigation-kernel candidates (policy.py:153-175) --------------------------------------------------
int omds_kernel_candidates(omds_ctx* ctx, float thr_dist, float thr_kernel, float thr_dot, const float* mu_c,
                           const float* sigma_c, int K, int cap, float* cand_q, int32_t* cand_th, int32_t* count) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(count && cap >= 0 && K >= 0 && K <= ctx->cfg.n_kernel_max, OMDS_ERR_INVALID_ARG,
            "omds_kernel_candidates: bad arguments");
    REQUIRE(K == 0 || (mu_c && sigma_c), OMDS_ERR_INVALID_ARG, "omds_kernel_candidates: null kernel means");
    REQUIRE(cap == 0 || (cand_q && cand_th), OMDS_ERR_INVALID_ARG, "omds_kernel_candidates: null output with cap > 0");
    CK(hipSetDevice(ctx->dev));
    const int N = ctx->cfg.n_traj, H = ctx->cfg.horizon, n = ctx->cfg.n_dof;
    const size_t need = (size_t)cap * n * 4 + (size_t)cap * 8;
    REQUIRE(need <= ctx->stage_bytes, OMDS_ERR_INVALID_ARG, "omds_kernel_candidates: cap too large for the staging buffer (<= N*H)");
    if (K > 0) {
        std::vector<float> means((size_t)K * (n + 1));
        std::memcpy(means.data(), mu_c, (size_t)K * n * 4);
        std::memcpy(means.data() + (size_t)K * n, sigma_c, (size_t)K * 4);
        CK(hipMemcpyAsync(ctx->d_means, means.data(), means.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
    }
    float* d_q = ctx->d_stage;
    int* d_th = reinterpret_cast<int*>(ctx->d_stage + (size_t)cap * n);
    omds_launch_candidates(ctx->stream, N, H, n, K, ctx->d_trajT, ctx->d_distT, ctx->d_dotT, ctx->d_means, thr_dist,
                           thr_kernel, thr_dot, ctx->prm.rbf_p, ctx->d_cflags, ctx->d_ccounts, ctx->d_coffsets, cap, d_q, d_th);
    CK(hipGetLastError());
    int32_t total = 0;
    CK(hipMemcpyAsync(&total, ctx->d_coffsets + N, 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    *count = total;
    const int m = std::min<int>(total, cap);
    if (m > 0) {
        CK(hipMemcpyAsync(cand_q, d_q, (size_t)m * n * 4, hipMemcpyDeviceToHost, ctx->stream));
        CK(hipMemcpyAsync(cand_th, d_th, (size_t)m * 8, hipMemcpyDeviceToHost, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
    }
    return OMDS_OK;
}

// ---- screening controls ------------------------------------------------------------------------------
int omds_set_screening(omds_ctx* ctx, int mode, float eps) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(mode >= -1 && mode <= 2 && eps == eps, OMDS_ERR_INVALID_ARG, "omds_set_screening: mode in {-1, 0, 1, 2}, eps not NaN");
    ctx->screen_mode = mode;
    if (eps > 0.f) {            // the caller's bound instead of a calibration (the run-time checks still widen it when they must)
        ctx->screen_eps = eps; ctx->screen_eps_fixed = true; ctx->screen_cal = true;
        ctx->obs_cal = ctx->obs_now;
        ctx->screen_suspended = false; ctx->screen_consec = 0;
    } else if (eps < 0.f) {     // forget the calibration: measured again at the next screened propagate
        ctx->screen_eps = 0.f; ctx->screen_eps_fixed = false; ctx->screen_cal = false;
        ctx->screen_suspended = false; ctx->screen_consec = 0;
    }                           // eps == 0: the mode only; bound, calibration and everything measured so far stay
    return OMDS_OK;
}
int omds_set_screening_audit(omds_ctx* ctx, int one_in) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(one_in >= 0 && one_in <= (1 << 20) && (one_in & (one_in - 1)) == 0, OMDS_ERR_INVALID_ARG,
            "omds_set_screening_audit: one_in must be 0 (no audit rows) or a power of two <= 2^20");
    ctx->audit_one_in = one_in;
    return OMDS_OK;
}
int omds_set_screening_sweep(omds_ctx* ctx, int every, int all_steps) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(every >= 0 && (all_steps == 0 || all_steps == 1), OMDS_ERR_INVALID_ARG, "omds_set_screening_sweep: every >= 0 (0 = no sweeps), all_steps in {0, 1}");
    ctx->sweep_every = every;
    ctx->sweep_all_steps = all_steps != 0;
    return OMDS_OK;
}
int omds_screen_sweep_hist(omds_ctx* ctx, uint64_t* words, int n_words, int reset) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(words && n_words == OMDS_SWEEP_HIST_WORDS, OMDS_ERR_INVALID_ARG, "omds_screen_sweep_hist: words must hold OMDS_SWEEP_HIST_WORDS entries");
    CK(hipSetDevice(ctx->dev));
    std::memset(words, 0, (size_t)n_words * 8);
    if (!ctx->d_sweep_hist) return OMDS_OK;   // no sweep has run yet
    CK(hipStreamSynchronize(ctx->stream));
    CK(hipMemcpy(words, ctx->d_sweep_hist, (size_t)n_words * 8, hipMemcpyDeviceToHost));
    words[OMDS_HIST_STEPS] = (uint64_t)ctx->screen_sweeps;
    if (reset) { CK(hipMemset(ctx->d_sweep_hist, 0, (size_t)n_words * 8)); ctx->screen_sweeps = 0; }
    return OMDS_OK;
}
int omds_screen_sweep_stats(omds_ctx* ctx, int32_t* every, int64_t* sweeps, float* sweep_max_err) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    if (every) *every = ctx->sweep_every;
    if (sweeps) *sweeps = ctx->screen_sweeps;
    if (sweep_max_err) *sweep_max_err = ctx->screen_sweep_err_seen;
    return OMDS_OK;
}
int omds_screen_order_stats(omds_ctx* ctx, int64_t* reorders, int32_t* never_fired, int n_levels) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(n_levels >= 0 && n_levels <= OMDS_MAX_HIDDEN + 1, OMDS_ERR_INVALID_ARG, "omds_screen_order_stats: 0 <= n_levels <= 9");
    if (reorders) *reorders = ctx->scr_reorders;
    if (never_fired)
        for (int L = 0; L < n_levels; ++L) never_fired[L] = ctx->scr_never_fired[L];
    return OMDS_OK;
}
// the exact zero-skip of k_pass1 (omds.h): firing units and multiplied chunks per tile and hidden level, from the device's counters
int omds_pass1_skip_stats(omds_ctx* ctx, int32_t* active, double* mean_units, double* mean_chunks, int n_levels, int64_t* tiles) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(ctx->have_mlp, OMDS_ERR_NOT_INITIALISED, "distance network not set (omds_set_mlp)");
    REQUIRE(n_levels >= 0 && n_levels <= OMDS_MAX_HIDDEN + 1, OMDS_ERR_INVALID_ARG, "omds_pass1_skip_stats: 0 <= n_levels <= 9");
    const bool on = ctx->mlp.compact && !ctx->wide.on;
    if (active) *active = on ? 1 : 0;
    unsigned long long h[2 * (OMDS_MAX_HIDDEN + 1) + 2] = {0};
    if (on && ctx->mlp.skip_stats) {
        CK(hipSetDevice(ctx->dev));
        CK(hipStreamSynchronize(ctx->stream));
        CK(hipMemcpy(h, ctx->mlp.skip_stats, sizeof(h), hipMemcpyDeviceToHost));
    }
    const double nt = h[0] ? (double)h[0] : 1.0;
    for (int L = 0; L < n_levels; ++L) {
        if (mean_chunks) mean_chunks[L] = on && L <= ctx->mlp.nhh ? (double)h[1 + L] / nt : 0.0;
        if (mean_units) mean_units[L] = on && L <= ctx->mlp.nhh ? (double)h[1 + (OMDS_MAX_HIDDEN + 1) + L] / nt : 0.0;
    }
    if (tiles) *tiles = (int64_t)h[0];
    return OMDS_OK;
}
int omds_screen_fallback_stats(omds_ctx* ctx, int64_t* by_error, int64_t* by_slack, int64_t* by_overflow, int64_t* suspensions) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    if (by_error) *by_error = ctx->screen_fb_error;
    if (by_slack) *by_slack = ctx->screen_fb_slack;
    if (by_overflow) *by_overflow = ctx->screen_fb_overflow;
    if (suspensions) *suspensions = ctx->screen_suspensions;
    return OMDS_OK;
}
int omds_screen_audit_stats(omds_ctx* ctx, int32_t* one_in, double* audit_rows_per_rollout_step, float* audit_max_err,
                            int32_t* suspended, int64_t* calibrations) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    if (one_in) *one_in = ctx->audit_one_in;
    if (audit_rows_per_rollout_step) *audit_rows_per_rollout_step = ctx->screen_steps > 0 ? ctx->screen_audit_rows / ctx->screen_steps : 0.0;
    if (audit_max_err) *audit_max_err = ctx->screen_audit_err_seen;
    if (suspended) *suspended = ctx->screen_suspended ? 1 : 0;
    if (calibrations) *calibrations = ctx->screen_recals;
    return OMDS_OK;
}
#ifdef OMDS_TEST_HOOKS
// Test hooks (include/omds_test.h; libomds_hip_test.so only -- the release library does not export them).
// omds_screen_debug_corrupt (tests/test_gpu_screen_audit.py): damages the screening network's inputs so that the run-time
// checks have something to catch.  what = 0: zeroes weight fragment `index` (1 KiB of slice index / 16) of the fp16 pack -- every
// screening value moves; what = 1: shifts obstacle `index` by `value` along x in the SCREENING input table only (undone by
// the next omds_set_obstacles) -- the fp16 network sees that one sphere elsewhere, so only the audit rows can notice.
int omds_debug_force_tile_rows(int tail_sel_rows, int tail_rows) {
    if (!((tail_sel_rows == 0 || tail_sel_rows == 4 || tail_sel_rows == 16 || tail_sel_rows == 32) && (tail_rows == 0 || tail_rows == 4 || tail_rows == 16 || tail_rows == 32)))
        return OMDS_ERR_INVALID_ARG;
    omds_force_tile_rows(tail_sel_rows, tail_rows);
    return OMDS_OK;
}

int omds_screen_debug_corrupt(omds_ctx* ctx, int what, int index, float value) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(ctx->screen_ok, OMDS_ERR_UNSUPPORTED, "omds_screen_debug_corrupt: no screening network for this model");
    CK(hipSetDevice(ctx->dev));
    CK(hipStreamSynchronize(ctx->stream));
    if (what == 0) {
        const int nfrag = (ctx->mlp.nhh * 8 + 2) * 16;
        REQUIRE(index >= 0 && index < nfrag, OMDS_ERR_INVALID_ARG, "omds_screen_debug_corrupt: fragment index out of range");
        CK(hipMemset(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(ctx->screen.Wh)) + (size_t)index * 1024, 0, 1024));
        return OMDS_OK;
    }
    REQUIRE(what == 1 && index >= 0 && index < ctx->n_obs, OMDS_ERR_INVALID_ARG, "omds_screen_debug_corrupt: what in {0, 1}, obstacle index in range");
    const int n = ctx->cfg.n_dof, d = ctx->mlp.d, ld = ctx->cfg.max_obs;
    const float x = ctx->obs_now[(size_t)index * 4] + value;
    const float f[3] = {x, std::sin(x), std::cos(x)};
    for (int part = 0; part < 3; ++part) {
        const uint16_t h = f32_to_f16_bits(f[part]);
        CK(hipMemcpy(ctx->d_FpH + omds_screen_fidx(part * d + n, index, ld), &h, 2, hipMemcpyHostToDevice));
    }
    return OMDS_OK;
}
#endif   // OMDS_TEST_HOOKS
// Diagnostic: the screening network alone on a batch (what k_select sees), for tests and for measuring eps.
int omds_screen_mindist(omds_ctx* ctx, const float* q, int B, float* mindist) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(q && mindist && B >= 1 && B <= ctx->cfg.n_traj, OMDS_ERR_INVALID_ARG, "omds_screen_mindist: need 1 <= batch <= n_traj and non-null arrays");
    int rc;
    if ((rc = check_ready(ctx, false))) return rc;
    REQUIRE(ctx->screen_ok, OMDS_ERR_UNSUPPORTED, "omds_screen_mindist: no screening network for this model (ReLU, 2..5 hidden layers)");
    CK(hipSetDevice(ctx->dev));
    const int n = ctx->cfg.n_dof, O = ctx->n_obs;
    CK(hipMemcpyAsync(ctx->d_stage, q, (size_t)B * n * 4, hipMemcpyHostToDevice, ctx->stream));
    omds_launch_transpose(ctx->stream, ctx->d_stage, ctx->d_qstage, B, n);
    omds_launch_rollout_features(ctx->stream, ctx->mlp, ctx->d_qstage, B, B, ctx->d_Fq, ctx->d_FqH, ctx->cfg.n_traj);
    omds_launch_screen(ctx->stream, ctx->screen, ctx->mlp, ctx->d_FqH, ctx->cfg.n_traj, ctx->d_FpH, ctx->cfg.max_obs, ctx->d_radius, O, B, ctx->prm.ignored_links, ctx->d_Dmin);
    CK(hipGetLastError());
    CK(hipMemcpyAsync(mindist, ctx->d_Dmin, (size_t)B * O * 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    return OMDS_OK;
}
int omds_screen_stats(omds_ctx* ctx, int32_t* active, float* eps, float* max_err_seen, double* cand_per_rollout_step,
                      int64_t* fallbacks) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    if (active) *active = (ctx->screen_ok && screen_wanted(ctx)) ? 1 : 0;
    if (eps) *eps = ctx->screen_eps;
    if (max_err_seen) *max_err_seen = ctx->screen_err_seen;
    if (cand_per_rollout_step) *cand_per_rollout_step = ctx->screen_steps > 0 ? ctx->screen_rows / ctx->screen_steps : 0.0;
    if (fallbacks) *fallbacks = ctx->screen_fallbacks;
    return OMDS_OK;
}

// ---- measurement -------------------------------------------------------------------------------------
int omds_prof_enable(omds_ctx* ctx, int on) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    ctx->prof_on = on != 0;
    ctx->prof_stride = on > 1 ? on : 1;
    return OMDS_OK;
}
int omds_prof_reset(omds_ctx* ctx) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    ctx->prof.ms = 0.0;
    ctx->prof_seen = 0;
    ctx->prof.launches = 0;
    ctx->prof.rows = 0;
    ctx->prof.flops = 0.0;
    ctx->prof.used = 0;
    ctx->screen_rows = 0.0;
    ctx->screen_steps = 0.0;
    ctx->screen_audit_rows = 0.0;
    return OMDS_OK;
}
int omds_prof_read_ex(omds_ctx* ctx, double* ms, int64_t* launches, double* flops, const char** kernel) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    { const int rc = prof_collect(ctx); if (rc) return rc; }
    if (ms) *ms = ctx->prof.ms;
    if (launches) *launches = ctx->prof.launches;
    if (flops) *flops = ctx->prof.flops;
    if (kernel) *kernel = ctx->prof.kernel;
    return OMDS_OK;
}
int omds_prof_read(omds_ctx* ctx, double* pass1_ms, int64_t* pass1_launches, int64_t* pass1_rows) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    { const int rc = prof_collect(ctx); if (rc) return rc; }
    if (pass1_ms) *pass1_ms = ctx->prof.ms;
    if (pass1_launches) *pass1_launches = ctx->prof.launches;
    if (pass1_rows) *pass1_rows = ctx->prof.rows;
    return OMDS_OK;
}

}  // extern "C"
