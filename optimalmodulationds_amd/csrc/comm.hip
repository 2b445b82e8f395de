// Multi-GPU exchange of the cost-weighted update: rollouts shard over one process per GPU, and the only
// data that crosses GPUs is what MPPI.shift_policy_means / get_qdot reduce over ALL rollouts (MPPI.py:319-345,
// policy.py:88-113).  Everything runs on the context's HIP stream on device buffers -- no host bounce:
//
//   k_cost_sum  -> d_red2 = [sum cost, N_local]        ncclAllReduce SUM (8 bytes)      -> global beta
//   k_weights, k_policy_sums -> d_red[0 : n_sum]       ncclAllReduce SUM (<= 3.4 KB)    -> update of mu/sigma/alpha
//   d_red[n_sum : n_sum+1+n] = (min cost, its qdot)    ncclAllGather (only for 'best')  -> MINLOC on the host
//
// then ONE D2H copy + stream sync and the O(K n) host arithmetic of omds_apply_update.  RCCL (xGMI on the box) is
// dlopen'ed the first time a communicator is asked for: single-GPU processes never load it.  The copy beside the loaded
// HIP runtime is used (in a process that imported PyTorch first that is PyTorch's own, already mapped).
#include <cstdlib>
#include <cstring>
#include <new>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include "omds_internal.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    std::string err;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    Rccl() {
        // The RCCL that belongs to the HIP runtime THIS library is bound to comes first: the one in the directory of the
        // loaded libamdhip64 (a PyTorch wheel bundles its own HIP + HSA + RCCL; which HIP a process ends up with depends
        // on whether torch or this library was loaded first, and an RCCL from the other set opens a second, uninitialised
        // HSA runtime: "no ROCm-capable device is detected" at ncclCommInitRank).  If that is the copy the process already
        // has mapped, dlopen returns the same handle -- no second RCCL.
        // OMDS_RCCL_LIB: use exactly this library (deployments with several ROCm installs; the test of the failure path)
        if (const char* forced = getenv("OMDS_RCCL_LIB")) {
            handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!handle) {
                const char* e = dlerror();   // ONE call: dlerror() clears the message it returns
                err = std::string("RCCL not available (OMDS_RCCL_LIB=") + forced + "): " + (e ? e : "dlopen failed");
                return;
            }
        }
        Dl_info info{};
        if (!handle && dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                for (const char* nm : {"librccl.so.1", "librccl.so"}) {
                    handle = dlopen((dir + nm).c_str(), RTLD_NOW | RTLD_LOCAL);
                    if (handle) break;
                }
            }
        }
        if (!handle) {   // then a copy that is already mapped (RTLD_NOLOAD), then the default search and the ROCm install
            for (const char* nm : {"librccl.so.1", "librccl.so"}) {
                handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
                if (handle) break;
            }
        }
        if (!handle) {
            for (const char* nm : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
                handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
                if (handle) break;
            }
        }
        if (!handle) {
            const char* e = dlerror();       // ONE call: the second would return NULL (std::string + nullptr is undefined)
            err = std::string("RCCL not available: ") + (e ? e : "dlopen failed");
            return;
        }
#define OMDS_RCCL_SYM(f)                                                                   \
    f = reinterpret_cast<decltype(f)>(dlsym(handle, "nccl" #f));                           \
    if (!f) { err = "RCCL symbol nccl" #f " missing"; handle = nullptr; return; }
        OMDS_RCCL_SYM(GetUniqueId)
        OMDS_RCCL_SYM(CommInitRank)
        OMDS_RCCL_SYM(CommDestroy)
        OMDS_RCCL_SYM(AllReduce)
        OMDS_RCCL_SYM(AllGather)
        OMDS_RCCL_SYM(GetErrorString)
#undef OMDS_RCCL_SYM
    }
    bool ok() const { return handle != nullptr; }
};

Rccl& rccl() { static Rccl r; return r; }
thread_local std::string g_comm_err;

}  // namespace

#define CK(expr) OMDS_HIP_CHECK(ctx, expr)
#define REQUIRE(cond, code, msg) do { if (!(cond)) { ctx->err = (msg); return (code); } } while (0)
#define CKN(expr)                                                                                     \
    do {                                                                                              \
        ncclResult_t _r = (expr);                                                                     \
        if (_r != ncclSuccess) {                                                                      \
            ctx->err = std::string(#expr) + ": " + rccl().GetErrorString(_r);                         \
            return OMDS_ERR_RCCL;                                                                     \
        }                                                                                             \
    } while (0)

void omds_comm_release(omds_ctx* ctx) {
    if (ctx->comm) { (void)rccl().CommDestroy(static_cast<ncclComm_t>(ctx->comm)); ctx->comm = nullptr; }
    if (ctx->d_gather) { (void)hipFree(ctx->d_gather); ctx->d_gather = nullptr; }
    if (ctx->h_gather) { (void)hipHostFree(ctx->h_gather); ctx->h_gather = nullptr; }
    ctx->comm_rank = 0;
    ctx->comm_world = 1;
}

extern "C" {

const char* omds_comm_last_error(void) { return g_comm_err.c_str(); }

int omds_comm_probe(void) {
    if (!rccl().ok()) { g_comm_err = rccl().err; return OMDS_ERR_RCCL; }
    return OMDS_OK;
}

int omds_comm_unique_id(uint8_t* out128) {
    if (!out128) { g_comm_err = "omds_comm_unique_id: null output"; return OMDS_ERR_INVALID_ARG; }
    if (!rccl().ok()) { g_comm_err = rccl().err; return OMDS_ERR_RCCL; }
    static_assert(OMDS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    const ncclResult_t r = rccl().GetUniqueId(&id);
    if (r != ncclSuccess) { g_comm_err = std::string("ncclGetUniqueId: ") + rccl().GetErrorString(r); return OMDS_ERR_RCCL; }
    std::memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return OMDS_OK;
}

int omds_comm_init_rank(omds_ctx* ctx, const uint8_t* id128, int rank, int world) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    REQUIRE(id128 && world >= 1 && rank >= 0 && rank < world, OMDS_ERR_INVALID_ARG,
            "omds_comm_init_rank: need a 128-byte id and 0 <= rank < world");
    REQUIRE(rccl().ok(), OMDS_ERR_RCCL, rccl().err);
    CK(hipSetDevice(ctx->dev));
    omds_comm_release(ctx);
    ncclUniqueId id;
    std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    CKN(rccl().CommInitRank(&comm, world, id, rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    const size_t gb = (size_t)world * (1 + OMDS_MAX_DOF) * sizeof(float);
    CK(hipMalloc(&ctx->d_gather, gb));
    CK(hipHostMalloc(&ctx->h_gather, gb));
    return OMDS_OK;
}

int omds_comm_destroy(omds_ctx* ctx) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    CK(hipSetDevice(ctx->dev));
    CK(hipStreamSynchronize(ctx->stream));
    omds_comm_release(ctx);
    return OMDS_OK;
}

int omds_comm_active(const omds_ctx* ctx) { return (ctx && ctx->comm) ? 1 : 0; }

int omds_comm_info(const omds_ctx* ctx, int32_t* rank, int32_t* world) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    if (rank) *rank = ctx->comm_rank;
    if (world) *world = ctx->comm ? ctx->comm_world : 1;
    return OMDS_OK;
}

// MPPI.shift_policy_means + get_qdot over ALL shards.  Without a communicator this is the single-shard update (the same
// kernels, the same summation order as omds_weighted_update).
int omds_weighted_update_sharded(omds_ctx* ctx, float rate, float ker_thr, float* mu_c, float* sigma_c, float* alpha_c,
                                 int32_t* mask_out, float* qdot_weighted, float* qdot_best, float* n_total_out) {
    return omds_update_impl(ctx, true, rate, ker_thr, mu_c, sigma_c, alpha_c, mask_out, qdot_weighted, qdot_best, n_total_out);
}

}  // extern "C"

int omds_update_impl(omds_ctx* ctx, bool use_comm, float rate, float ker_thr, float* mu_c, float* sigma_c, float* alpha_c,
                     int32_t* mask_out, float* qdot_weighted, float* qdot_best, float* n_total_out) {
    if (!ctx) return OMDS_ERR_INVALID_ARG;
    const int N = ctx->cfg.n_traj, n = ctx->cfg.n_dof, K = ctx->n_kernels, H = ctx->cfg.horizon;
    REQUIRE(K == 0 || (mu_c && sigma_c && alpha_c), OMDS_ERR_INVALID_ARG, "omds_weighted_update_sharded: null mean array");
    REQUIRE(ctx->have_cost_vals, OMDS_ERR_NOT_INITIALISED, "no cost available: call omds_cost after omds_propagate");
    CK(hipSetDevice(ctx->dev));
    ncclComm_t comm = use_comm ? static_cast<ncclComm_t>(ctx->comm) : nullptr;
    const int world = comm ? ctx->comm_world : 1;
    const int rs = omds_red_size(K, n), n_sum = rs - (1 + n);
    float* red2 = ctx->d_red + rs;   // [sum cost, N] lives behind the packed buffer
    const bool gather = comm && qdot_best && world > 1;
    auto enqueue = [&]() -> int {
        omds_launch_cost_sum(ctx->stream, ctx->d_cost, N, red2);
        if (comm) CKN(rccl().AllReduce(red2, red2, 2, ncclFloat, ncclSum, comm, ctx->stream));
        omds_launch_weights(ctx->stream, ctx->d_cost, N, red2, ctx->d_w, nullptr);
        omds_launch_policy_sums(ctx->stream, N, n, K, ctx->d_w, ctx->d_muT, ctx->d_sigmaT, ctx->d_alphaT, ctx->d_maxact,
                                ctx->d_phisum0, ctx->d_qdotT, ctx->d_cost, (!comm || ctx->comm_rank == 0) ? 1 : 0, ctx->d_red);
        CK(hipGetLastError());
        if (comm) CKN(rccl().AllReduce(ctx->d_red, ctx->d_red, (size_t)n_sum, ncclFloat, ncclSum, comm, ctx->stream));
        if (gather) {
            CKN(rccl().AllGather(ctx->d_red + n_sum, ctx->d_gather, (size_t)(1 + n), ncclFloat, comm, ctx->stream));
            CK(hipMemcpyAsync(ctx->h_gather, ctx->d_gather, (size_t)world * (1 + n) * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        CK(hipMemcpyAsync(ctx->h_red, ctx->d_red, (size_t)(rs + 2) * 4, hipMemcpyDeviceToHost, ctx->stream));
        return OMDS_OK;
    };
    { const int erc = enqueue(); if (erc) return erc; }
    CK(hipStreamSynchronize(ctx->stream));
    const float* red = ctx->h_red;
    const float n_total = red[rs + 1];
    if (n_total_out) *n_total_out = n_total;
    const int rc = omds_apply_update(K, n, H, red, n_total, rate, ker_thr, ctx->prm.variant, mu_c, sigma_c, alpha_c, mask_out);
    if (rc) { ctx->err = "omds_apply_update: invalid argument"; return rc; }
    const int o_qd = 1 + K * (2 * n + 3);
    if (qdot_weighted)
        for (int j = 0; j < n; ++j) qdot_weighted[j] = red[o_qd + j] / red[0];
    if (qdot_best) {
        const float* best = red + n_sum;   // this shard's (min cost, qdot)
        if (gather) {                      // lowest rank on ties, like a global arg-min over rank-ordered rollouts
            for (int r = 0; r < world; ++r) {
                const float* g = ctx->h_gather + (size_t)r * (1 + n);
                if (r == 0 || g[0] < best[0]) best = g;
            }
        }
        for (int j = 0; j < n; ++j) qdot_best[j] = best[1 + j];
    }
    return OMDS_OK;
}
