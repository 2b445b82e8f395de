// Internal declarations shared by the HIP translation units of libomds_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <atomic>
#include <vector>

#include "omds.h"

// Experiment knobs.  Environment variables that change a kernel choice or a tile shape, switch a run-time guard off or make a
// kernel return early (timing experiments: wrong results by design) exist only in builds with -DOMDS_EXPERIMENT (`make experiment`,
// `make variant`, `make timeline`: libraries the package loads only when OMDS_LIB names them).  The release library reads
// OMDS_SCREEN (0 | 1: the screening mode of contexts left on "auto"), OMDS_ROCTX and OMDS_RCCL_LIB, nothing else.
#ifdef OMDS_EXPERIMENT
#include <cstdlib>
#define OMDS_EXP_ENV(name, dflt) ([&]() -> int { const char* _e = getenv(name); return _e ? atoi(_e) : (dflt); }())
#define OMDS_DBG(x) (x)
#else
#define OMDS_EXP_ENV(name, dflt) (dflt)
#define OMDS_DBG(x) 0
#endif

constexpr int OMDS_WIDTH = 256;        // hidden width the MFMA kernels are specialised for
constexpr int OMDS_LDH = 260;          // LDS row stride of the activation tile (floats): 256 + 4 pad
constexpr int OMDS_CPAD = 16;          // output channels padded to one 16-wide MFMA tile
constexpr int OMDS_MAX_HIDDEN = 8;     // hidden layers supported (reference nets: 4)
constexpr int OMDS_NCB = OMDS_WIDTH / 32;  // 32-column blocks per hidden layer
constexpr int OMDS_FROW = 32;          // floats per row of the encoded-input tables Fq / Fp (3 (n_dof + 3) <= 30 features, zero-padded)

// ---- the reference's summation order on the MFMA --------------------------------------------------------------------------
// torch-CPU computes every Linear layer (network_macros_mod.py:137-146: addmm = MKL sgemm) and every product of the vjp
// (robot_sdf.py:153-158) as ONE fmaf chain per output element over k in ASCENDING order, starting from zero, the bias added
// afterwards (tools/studies/assoc_order_study.py: bit for bit for M >= 11 rows).  An fp32 MFMA is an fmaf chain over its K
// indices in lane-group order (tools/ubench/mfma_order.hip), so the order a kernel sums in is the order in which its A / B
// fragments present k.  The GEMM cores read the activation tile from LDS as 16-byte fragments: lane half h of the 32x32x2
// shape takes four consecutive floats at 8c + 4h and feeds them to four MFMAs, which therefore contract positions
// 8c + {0, 4, 1, 5, 2, 6, 3, 7} in that order (gemm16 and gemm4 visit the same sequence).  For that sequence to be k = 8c + 0..7
// the tile is STORED k-permuted inside every group of eight columns -- logical column k sits at position omds_kpos(k) -- and
// the weight packs put W[.][omds_kat(s)] where a fragment reads position s.  Every writer of an activation / gradient tile
// goes through omds_kpos; the GEMM cores are untouched.
__host__ __device__ inline int omds_kpos(int k) { return (k & ~7) | ((k & 1) << 2) | ((k & 7) >> 1); }   // column -> position
__host__ __device__ inline int omds_kat(int s) { return (s & ~7) | ((s & 3) << 1) | ((s >> 2) & 1); }    // position -> column

// Device-side view of the distance network, weights pre-packed into MFMA fragment order.
struct MlpDev {
    const float4* Wf;    // [nhh][8 colblk][32 kchunk][64 lane] forward pack of hidden->hidden layers
    const float4* Wb;    // same shape, transposed pack for the backward pass
    const float* bh;     // [nhh][256] hidden->hidden biases
    const float4* Wl;    // [16 kchunk][64 lane] last layer, 16x16x4 B-fragments (channels padded to 16)
    const float* bl;     // [16]
    const float* Wlraw;  // [C][256] last layer, row-major (backward seed)
    const float* Whraw;  // [nhh][256][256] hidden->hidden layers, row-major [out][in] (the small-O step's 4-row backward streams them)
    const float* W1t;    // [3d][256] first layer transposed (the small-O step's first-layer backward)
    const float* b1;     // [256]
    const float4* W1f;   // [8 colblk][4 kchunk][64 lane] first layer, forward pack over the 3d encoded inputs (padded to K = 32)
    const float4* W1f16; // [16 colblk16][2 kchunk][64 lane] the same for the 16-row tiles
    const float4* W1b;   // [32 kchunk][64 lane] first layer, backward pack (cols = 3d features, padded to 32)
    // EXACT ZERO-SKIP of k_pass1 by per-tile compaction (pass1_tile_dyn, mlp_device.h; ReLU networks without skips)
    const float* WhT;    // [nhh][256 unit][256 column] hidden->hidden weights transposed (a W^T row = the weights leaving one unit): the
                         // compacted products fetch their fragments by unit
    const float* WlT;    // [256 unit][16] last layer likewise (channels padded to 16)
    unsigned long long* skip_stats;   // device counters since omds_set_mlp: [0] tiles, [1 + L] sum over tiles of the chunks multiplied over
                                      // level L (of 8 positions; the last hidden level: of 16), [10 + L] sum of the firing units
    uint8_t compact;     // 1: k_pass1 runs pass1_tile_dyn (0: the network does not qualify, or OMDS_FLAG_DENSE_PASS1)
    // the same three packs in v_mfma_f32_16x16x4 fragment order, for the 16-row pass-2 tiles of small batches:
    // lane l of chunk c holds 4 consecutive k = 16c + 4(l>>4) .. +3 of column 16*cb + (l&15)
    const float4* Wf16;  // [nhh][16 colblk16][16 kchunk][64 lane]
    const float4* Wb16;  // same shape, transposed
    const float4* W1b16; // [16 kchunk][2 colblk16][64 lane]
    const float4* Wb4;   // [nhh][4 colblk64][64 kq][64 lane] backward pack for the 4-row-group GEMM (gemm4): lane l = W[4kq .. +3][64cb + l]
    const float4* Wf4;   // same shape, forward: lane l = W[64cb + l][positions 4kq .. +3] (pass2_body_g4: the all-fp32 tail on 4-row groups)
    int nhh;             // number of hidden->hidden layers (= hidden layers - 1)
    int C;               // output channels (links)
    int d;               // raw inputs: n_dof + 3 (obstacle x, y, z), or n_dof + 2 for the toy networks (x, y)
    int n_dof;
    float out_div;
    int act;             // OMDS_ACT_RELU | OMDS_ACT_TANH
    // Skip-connection networks (network_macros_mod.py:117-146: the encoded input is concatenated behind the activations of a
    // hidden layer).  Level 0 = output of layer 1, level l + 1 = output of hidden->hidden layer l.  The layer in front of a
    // concatenation is narrower by 3d, so the concatenated vector still has <= 256 columns: the encoded input is written into
    // columns skip_col[L] .. +3d-1 of the activation tile, whose padded units are zero there.
    uint32_t skip_mask;  // bit L: concatenation behind level L
    uint8_t skip_col[OMDS_MAX_HIDDEN + 1];
    // the same encoded input as fp16 at the slots of the CONCATENATED columns of the screening kernel (omds_screen_sidx), written
    // beside FqH / FpH with their row capacities; null without skips or without a screening network
    uint16_t* scrQ;      // [4 pieces][n_traj][8]
    uint16_t* scrP;      // [4 pieces][max_obs][8]
#ifdef OMDS_TIMELINE
    unsigned long long* tl;   // diagnostic build only (make TIMELINE=1): [workgroup][8] phase timestamps of k_pass1
#endif
};

// fp16 screening network (screen_kernel.hip): hidden->hidden and last-layer weights as fp16 A-fragment slices
struct ScreenDev {
    const void* Wh = nullptr;     // [nhh*8 + 2 slices][16 fragments][64 lane][8 halfs]: layer 1, hidden->hidden, last layer
    const float* bias = nullptr;  // [nhh + 2][256]
};

// element index of network input feature kappa (of [x, sin x, cos x], < 32) of row t in the fp16 input tables of the
// screening kernel: 16-byte piece kappa / 8 (= k-chunk kappa / 16, lane-half (kappa / 8) & 1), slot kappa % 8
__host__ __device__ inline size_t omds_screen_fidx(int kappa, int t, int stride) {
    return ((size_t)(kappa >> 3) * stride + t) * 8 + (kappa & 7);
}

// The screening kernel's weight pack puts the 3d concatenated input columns of a layer behind a skip concatenation LAST: virtual
// columns 256 - 3d .. 255 = k-chunks 14 and 15 (the K order of a dot product is free).  Element index of feature kappa
// (< 3d <= 32) of row t in the skip tables: virtual column v -> chunk v / 16 - 14, C-layout slot as for every hidden activation
// (lane-half ((v & 15) >> 2) & 1, slot 4 ((v & 15) >> 3) + (v & 3)); 16-byte piece 2 chunk + half
__host__ __device__ inline size_t omds_screen_sidx(int kappa, int F, int t, int stride) {
    const int v = 256 - F + kappa, f = v & 15;
    const int piece = 2 * ((v >> 4) - 14) + ((f >> 2) & 1), slot = 4 * (f >> 3) + (f & 3);
    return ((size_t)piece * stride + t) * 8 + slot;
}

// hipFuncSetAttribute applies to the current device only: true the first time a kernel is launched on each device.
inline bool omds_first_use_on_device(std::atomic<uint64_t>& mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    return (mask.fetch_or(bit) & bit) == 0;
}

// A distance network with a hidden layer wider than OMDS_WIDTH (wide_kernels.hip): raw row-major weights, materialised activations
struct WideNet {
    bool on = false;
    std::vector<int> dims;            // [L + 1]: 3 d, hidden ..., C
    int d = 0, act = 0;
    float out_div = 1.f;
    std::vector<float*> W, b;         // device, torch layout [out][in]
    int chunk_rows = 0;               // pass-1 rows per chunk
    float* X = nullptr;               // [chunk_rows][3 d] encoded inputs of a pass-1 chunk
    float* H[2] = {nullptr, nullptr}; // [chunk_rows][max width] activation ping-pong of pass 1
    int rows2 = 0;                    // pass-2 row capacity (n_traj * n_closest)
    float* X2 = nullptr;              // [rows2][3 d]
    std::vector<float*> A;            // [L] activations of every layer of the pass-2 rows
    float* G[2] = {nullptr, nullptr}; // [rows2][max width] gradient ping-pong
};

struct ProfEvents {
    std::vector<hipEvent_t> start, stop;
    size_t used = 0;
    double ms = 0.0;
    int64_t launches = 0, rows = 0;
    double flops = 0.0;          // algorithmic FLOPs of the bracketed launches
    const char* kernel = "k_pass1";
};

__host__ __device__ inline int omds_seds_stride(int n) { return 2 * n + 2 + 2 * n * n; }
struct StepArgs {
    int N, H, n, K, Kmax, k, d, step;   // step = i in 1..H
    float* trajT; float* distT; float* dotT; float* actT; float* normalT; float* kvalT; float* qdotT;
    float* maxact; float* phisum0;
    const float* muT; const float* sigmaT; const float* alphaT;
    const float* gradx; const float* drow;
    float qf[OMDS_MAX_DOF];
    const float* A;   // [n][n] nominal DS matrix of MPPI_toy (velocity = (q - qf) @ A), nullptr = LinDS
    const float* seds;   // SEDS components [G][omds_seds_stride(n)]: mu_in[n], b[n], prior, den, sigma_inv[n][n], A[n][n]; nullptr = not SEDS
    int seds_G;
    float seds_lin_thr, seds_thr;
    omds_params prm;
};

struct omds_ctx {
    omds_config cfg{};
    omds_params prm{};
    int dev = 0;
    hipStream_t stream = nullptr;
    std::string err;
    float* h_verdict = nullptr;          // pinned [4 + H + 2]: d_scerr, d_sctotal of the propagate being finished
    // network
    bool have_mlp = false;
    MlpDev mlp{};
    WideNet wide{};              // networks with a hidden layer wider than 256 (its buffers live in mlp_allocs)
    std::vector<void*> mlp_allocs;
    int act = OMDS_ACT_RELU;
    double f_fwd = 0.0, f_bwd = 0.0;   // algorithmic FLOPs of one network forward / backward row
    // screening (pass 1 in fp16 + exact re-selection, screen_kernel.hip)
    ScreenDev screen{};
    bool screen_ok = false;      // packs present (ReLU network)
    int screen_mode = -1;        // -1 = auto (on for large pair counts), 0 = off, 1 = forced on
    float screen_eps = 0.f;      // bound assumed on |screening value - fp32 value| of the rows that are not re-evaluated; 0 = not calibrated yet
    bool screen_cal = false;
    int* d_rowlist = nullptr;    // [N*max_obs] candidate pairs
    int* d_range = nullptr;      // [N][2] each rollout's range of the list
    int ex_cap = 0;              // entries the k_exact output arrays hold (N * 32)
    float* d_exD = nullptr;      // [ex_cap] pass-1 value / pass-2 distance / arg-min link / ReLU masks of each list entry
    float* d_exDr = nullptr;
    int* d_exMin = nullptr;
    uint32_t* d_exMask = nullptr;
    // the all-fp32 step without a second forward (pass1_tile mode 6): per PAIR what k_exact leaves per candidate
    float* d_allDr = nullptr; int* d_allMin = nullptr; uint32_t* d_allMask = nullptr;   // [all_cap], [all_cap], [all_cap][all_nhid][8]
    long long all_cap = 0; int all_nhid = 0;
    long long all_failed_pairs = -1; int all_failed_nhid = -1;   // the last request these three could not be allocated for (not retried until it changes)
    float* d_exDeriv = nullptr;  // tanh networks: [hidden layers][ex_cap][256] activation derivatives of the list entries (allocated by omds_set_mlp)
    int* d_sctotal = nullptr;    // [H+2]: candidate rows listed per horizon step; [H+1]: audit entries recorded in this propagate
    int* d_audit_rows = nullptr; // [audit_cap] audit sample of a propagate: pair rows into d_FqAll's row space, their screening values
    float* d_audit_da = nullptr;
    int audit_cap = 0;
    float* d_FqAll = nullptr;  // [H][N][OMDS_FROW] encoded joint inputs at the states of every horizon step (kept by a screened propagate with an audit sample)
    unsigned* d_scerr = nullptr; // [4]: max |screening - exact| over the candidates (float bits); rollouts whose slack guard failed;
                                 //      max (screening - exact) over the audit sample (float bits); calibration scratch
    double screen_rows = 0.0;    // statistics since the last omds_prof_reset: candidate rows, (rollout, step)s, audit rows
    double screen_steps = 0.0;
    double screen_audit_rows = 0.0;
    long long screen_fallbacks = 0;      // since creation
    long long screen_fb_error = 0, screen_fb_slack = 0, screen_fb_overflow = 0;   // ... by what tripped them (omds_screen_fallback_stats)
    long long screen_suspensions = 0;    // times three fallbacks in a row (or a non-finite error) suspended screening
    long long screen_recals = 0;         // calibrations run since creation
    // Unit order of the screening pack (capi.hip: build_screen_pack / screen_reorder).  k_screen skips the k-chunks whose 16 units are
    // zero for all 32 pairs of a wave; which units fire depends on the trained weights (a third of the shipped network's never do),
    // so every calibration first sorts the hidden units by how often they fire on a uniform sample of its batch's pairs.
    std::vector<std::vector<float>> scr_W, scr_b;   // the network, zero-padded to width 256 (host copy for building the pack again)
    std::vector<int32_t> scr_out_dims;
    bool scr_reorder_pending = false;    // set by a calibration: the order is refined behind the next accepted propagate, on its rollouts' states
    long long scr_reorders = 0;
    int* d_scr_tmp = nullptr;            // [8] device words of screen_reorder: error words of its k_exact launch (unused), the list length
    int scr_never_fired[OMDS_MAX_HIDDEN + 1] = {0};   // per hidden level: units that fired in no row of the last reorder's sample
    float screen_err_seen = 0.f;         // largest |Da - D| seen on candidates since the last calibration
    float screen_audit_err_seen = 0.f;   // largest Da - D seen on audit rows since the last calibration
    bool screen_eps_fixed = false;       // eps given by the caller (omds_set_screening(mode, eps > 0)): never recalibrated
    bool screen_suspended = false;       // three propagates in a row fell back to fp32: the fp32 step until the next calibration
    int screen_consec = 0;               // consecutive fallbacks
    int audit_one_in = 128;              // a non-candidate pair is audited with probability 1 / audit_one_in (power of two; 0 = no audit)
    unsigned audit_counter = 0;          // feeds the audit hash: another sample every step of every propagate
    int sweep_every = 32;                // every sweep_every-th screened propagate checks ALL pairs of its last step in fp32 (0 = never)
    bool sweep_all_steps = false;        // ... of EVERY horizon step (soak runs: omds_set_screening_sweep(every, 1))
    float* d_sweepD = nullptr;           // [N*max_obs] fp32 values / screening values of the step being swept (allocated at the first sweep)
    float* d_sweepDa = nullptr;
    size_t sweep_cap = 0;                // pairs the two buffers hold
    unsigned long long* d_sweep_hist = nullptr;   // [OMDS_SWEEP_HIST_WORDS] accumulated statistics of every sweep since creation / the last reset
    long long screen_propagates = 0;     // screened propagates since creation
    bool sweep_force_next = false;       // the screening pack was re-sorted after its bound was measured: the next screened propagate carries a sweep
    long long screen_sweeps = 0;         // sweeps run since creation
    float screen_sweep_err_seen = 0.f;   // largest |Da - D| a sweep saw since the last calibration
    bool sweep_now = false;              // the propagate being finished carried a sweep (d_scerr[3] is valid)
    int sweep_steps_now = 0;             // ... of this many steps
    std::vector<float> obs_cal;          // the obstacle set the bound was calibrated against (omds_set_obstacles compares)
    std::vector<float> obs_now;          // host copy of the current obstacle set
    bool have_rollouts = false;          // d_trajT holds the rollouts of a finished propagate (calibration draws states from them)
    // scene
    int n_obs = 0;
    float* d_obs = nullptr;      // [max_obs][4]
    float* d_Fp = nullptr;     // [max_obs][OMDS_FROW] encoded obstacle points [p, sin p, cos p] at their feature slots (the joints' slots zero)
    uint16_t* d_FpH = nullptr;   // [4][n_obs][8] fp16 network inputs of the obstacle points for the screening kernel
    uint16_t* d_FqH = nullptr;   // [4][batch][8] rollout states likewise
    uint16_t* d_FqS = nullptr;   // skip-connection networks with a screening network: MlpDev::scrQ / scrP
    uint16_t* d_FpS = nullptr;
    float* d_listDa = nullptr;   // [N*max_obs] screening values of the candidate list (k_screen's selecting flush)
    void* d_sinks = nullptr;     // [H] SelectSink of every horizon step of the running propagate (device copy + pinned staging)
    void* h_sinks = nullptr;
    float* d_radius = nullptr;   // [max_obs]
    // DS / cost
    bool have_ds = false, have_cost = false;
    float qf[OMDS_MAX_DOF] = {0};
    float* d_A = nullptr;        // [n][n] MPPI_toy nominal DS matrix (omds_set_ds_matrix), used when have_A
    float* d_seds = nullptr;     // [G][omds_seds_stride(n)] SEDS components (omds_set_ds_seds), used when seds_G > 0
    int seds_G = 0;
    float seds_lin_thr = 1e-2f, seds_thr = 1e-2f;
    bool have_A = false;
    float qmin[OMDS_MAX_DOF] = {0}, qmax[OMDS_MAX_DOF] = {0};
    float dh[(OMDS_MAX_DOF + 1) * 4] = {0};
    float goal_fk[OMDS_MAX_DOF * 3] = {0};
    // rollout state, SoA (rollout index fastest)
    float* d_trajT = nullptr;    // [H][n][N]
    float* d_distT = nullptr;    // [H][N]
    float* d_dotT = nullptr;     // [H][N]
    float* d_actT = nullptr;     // [H][N]
    float* d_normalT = nullptr;  // [H][n][N]
    float* d_kvalT = nullptr;    // [H][Kmax][N]
    float* d_qdotT = nullptr;    // [n][N]
    float* d_maxact = nullptr;   // [Kmax][N] running max_h(phi*act)
    float* d_phisum0 = nullptr;  // [Kmax] sum_h phi of local rollout 0
    float* d_qstage = nullptr;   // [n][N] staging for dist_grad batches / per-rollout starts
    // policy samples, SoA
    int n_kernels = 0;
    float* d_muT = nullptr;      // [Kmax][n][N]
    float* d_sigmaT = nullptr;   // [Kmax][N]
    float* d_alphaT = nullptr;   // [Kmax][n][N]
    float* d_means = nullptr;    // [Kmax*(2n+1)] mu_c, sigma_c, alpha_c staging
    // network scratch
    float* d_Fq = nullptr;     // [Nrows][OMDS_FROW] encoded joint states [q, sin q, cos q] at their feature slots (d_Fq, d_Dmin and d_ex* are SCRATCH between propagates: calibration and the
    float* d_Dmin = nullptr;     // [N][max_obs]    screening pack's re-sort overwrite them after the results have been published)
    int32_t* d_idx = nullptr;    // [N][k]
    float* d_gradx = nullptr;    // [N*k][d]
    float* d_drow = nullptr;     // [N*k]
    float* d_yraw = nullptr;     // [N*k][16]
    int32_t* d_minidx = nullptr; // [N*k]
    float* d_dscr = nullptr;     // tanh only: [hidden layers][N*k padded to 32][256] activation derivatives
    float* d_dist = nullptr;     // [N]
    float* d_nngrad = nullptr;   // [N][n]
    float* d_uev = nullptr;      // omds_weighted_update_eval scratch (first use): cost [N], maxact [Kmax][N], phisum0 [Kmax], act [N][H]
    float* d_evalT = nullptr;    // omds_cost_eval scratch (first use): caller tensors in SoA + their cost
    float* d_vjp_xyzr = nullptr; // omds_mlp_forward_vjp scratch (first use): per-row points, their layer-1 halves, zero radii
    float* d_vjp_B = nullptr;
    int vjp_cap = 0;             // rows the three hold
    float* d_vjp_rad = nullptr;
    // cost / reduction
    float* d_cost = nullptr;     // [N]
    float* d_w = nullptr;        // [N] unnormalised weights
    float* d_red = nullptr;      // packed reduction buffer
    float* h_red = nullptr;      // pinned mirror
    float* h_in = nullptr;       // pinned staging of the small per-iteration inputs: [0, 7) q_cur, then the policy means
    float* d_qcur = nullptr;     // [n] start state of a broadcast propagate
    hipEvent_t ev_in_q = nullptr, ev_in_means = nullptr;   // the H2D copies out of h_in have executed (back-to-back calls)
    bool have_cost_vals = false;
    // kernel-candidate scratch
    unsigned char* d_cflags = nullptr;  // [N][H]
    int* d_ccounts = nullptr;           // [N]
    int* d_coffsets = nullptr;          // [N+1]
    // staging
    float* d_stage = nullptr;    // transposition staging for host copies
    size_t stage_bytes = 0;
    float* h_stage = nullptr;
    // multi-GPU (comm.hip): RCCL communicator of the rollout shards, nullptr = single shard
    void* comm = nullptr;        // ncclComm_t
    int comm_rank = 0, comm_world = 1;
    float* d_gather = nullptr;   // [world][1+n] (min cost, qdot of the arg-min) of every shard
    float* h_gather = nullptr;   // pinned mirror
    // profiling
    bool prof_on = false;
    int prof_stride = 1;         // bracket every prof_stride-th launch of the dominant kernel with events
    long long prof_seen = 0;     // launches of the dominant kernel since omds_prof_reset
    bool prof_open = false;      // prof_begin recorded a start event that prof_end has to close
    ProfEvents prof;
};

#define OMDS_HIP_CHECK(ctx, expr)                                                         \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);               \
            return OMDS_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

// comm.hip: releases the communicator and its buffers (called by omds_destroy)
void omds_comm_release(omds_ctx* ctx);
// the cost-weighted update on the context stream (one host sync); use_comm = reduce over the communicator's shards
int omds_update_impl(omds_ctx* ctx, bool use_comm, float rate, float ker_thr, float* mu_c, float* sigma_c, float* alpha_c,
                     int32_t* mask_out, float* qdot_weighted, float* qdot_best, float* n_total_out);

// ---- launchers implemented in mlp_kernels.hip ------------------------------------------------
// FqH / FpH (optional): the network inputs (q, sin q, cos q / the obstacle point likewise) as fp16 at their feature slots
// of the screening kernel's input tables, [4 pieces][ldF rows][8] (omds_screen_fidx).  ldF is the table's CAPACITY
// (n_traj / max_obs), never the batch: the slots the other operand owns must stay zero, and a batch-dependent stride would
// alias them with data of an earlier call
// slab > 0: the B rows are `B / slab` consecutive [n][ldq] state slabs of `slab` rollouts each (the stored rollouts trajT
// [H][n][N] as ONE batch of H*N states: row h*slab + t reads qT[(h*n + c) * ldq + t])
// Fq [B][OMDS_FROW] / Fp [O][OMDS_FROW]: the fp32 encoded inputs [x, sin x, cos x] of the rollout states / obstacle points at
// their feature slots (feature part * d + j; the other operand's slots and the padding stay zero: a pair's input row is the
// bitwise OR of its two rows).  The tables must have been zeroed once for the network's d (omds_set_mlp does)
void omds_launch_rollout_features(hipStream_t s, const MlpDev& m, const float* qT, int ldq, int B, float* Fq, uint16_t* FqH = nullptr, int ldF = 0,
                                  int slab = 0);
void omds_launch_obstacle_features(hipStream_t s, const MlpDev& m, const float* xyzr, int O, float* Fp, float* radius, uint16_t* FpH = nullptr, int ldF = 0);
void omds_launch_pass1(hipStream_t s, const MlpDev& m, const float* Fq, const float* Fp, const float* radius,
                       int O, int B, uint32_t ignored_links, float* Dmin);
void omds_launch_topk(hipStream_t s, const float* Dmin, int B, int O, int k, int32_t* idx);
void omds_launch_pass2(hipStream_t s, const MlpDev& m, const float* Fq, const float* Fp, const float* radius,
                       const float* xyzr, const int32_t* idx, int B, int k, const float* qT, int ldq,
                       float* gradx, float* drow, float* yraw, int32_t* minidx, float* dscr, int seed_col = -1);
void omds_launch_blend(hipStream_t s, const float* gradx, const float* drow, int B, int k, int d, int n,
                       float softmax_k, float* dist, float* nngrad);

// ---- train.hip: the trainer's exact-fp32 MFMA GEMM, for callers outside the trainer (wide_kernels.hip) -------------------
// Out [B][out] = act(H [B][in] . W^T + b), W [out][in] row-major like torch; act: OMDS_ACT_* or -1 (none)
void omds_launch_linear_forward(hipStream_t s, const float* H, int in, const float* W, const float* b, float* Out, int out, int B, int act);
// Gi [B][in] = (G [B][out] . W) * act'(Hact [B][in]); Hact == nullptr: no derivative factor
void omds_launch_linear_inputgrad(hipStream_t s, const float* G, int out, const float* W, int in, float* Gi, int B, const float* Hact, int act);
// ---- wide_kernels.hip ------------------------------------------------------------------------------
struct omds_ctx;
int omds_wide_network(omds_ctx* ctx, const float* qT, int ldq, int B);
int omds_wide_vjp(omds_ctx* ctx, const float* d_x, int rows, int seed_col = -1);

// ---- launchers implemented in screen_kernel.hip -----------------------------------------------
struct SelectSink;
// sel != nullptr (a DEVICE pointer to the step's sink; needs omds_screen_can_select(O)): every workgroup owns whole rollouts
// and its flush phase selects their candidates from LDS -- Dmin is not written and no k_select runs; otherwise the N x O
// matrix goes to Dmin
void omds_launch_screen(hipStream_t s, const ScreenDev& sd, const MlpDev& m, const uint16_t* FqH, int ldFq, const uint16_t* FpH,
                        int ldFp, const float* radius, int O, int B, uint32_t ignored, float* Dmin, const SelectSink* sel = nullptr);
bool omds_screen_supported(const MlpDev& m);
bool omds_screen_can_select(int O);
// What k_exact leaves behind for the screened step's tail (k_tail_sel), per entry of the candidate list: the pass-1 value,
// and everything pass 2's forward would produce for that row -- its arithmetic is the same bit for bit -- so that the tail
// only runs the backward: the pass-2 distance, the arg-min link, and the ReLU masks of every hidden layer.  mask layout per
// entry: [hidden layer][8 words], bit (col & 31) of word col >> 5 = the unit of (logical) column col fired.
// Window of the candidate list in units of the error bound eps: tau = (k-th smallest screening value) + OMDS_SCREEN_WINDOW * eps.
// eps bounds the screening error of the rows that are NOT re-evaluated; the extra quarter absorbs the shift of the k-th row
// itself (a re-evaluated row, whose error is measured: the slack guard of k_tail_sel checks tau - D*_k >= eps per rollout).
constexpr float OMDS_SCREEN_WINDOW = 1.25f;
// tanh screening networks: the pre-activations of every tanh layer are produced as 2 log2(e) x (fp16 weights and fp32 biases
// scaled by this factor when they are packed), so that the kernel's tanh is 1 - 2 / (1 + exp2(.)) without a multiply
constexpr float OMDS_SCREEN_TANH_SCALE = 2.885390081777927f;

struct ExactOut {
    float* D;          // [cap] pass-1 value
    float* dr;         // [cap] pass-2 distance y[argmin] / out_div - radius
    int* amin;         // [cap] arg-min link over all raw outputs
    uint32_t* mask;    // [cap][nhid][8]
    int cap;           // entries the arrays hold (the list may be longer: the host then redoes the propagate in fp32)
    const float* Da = nullptr;   // audit list only (pass1_tile mode 4): [entries] screening value of each listed pair
    float* deriv = nullptr;      // tanh networks (pass1_tile mode 5): [hidden layers][cap][256] 1 - h^2 of every entry's hidden units
};

void omds_launch_pass1_emit(hipStream_t s, const MlpDev& m, const float* Fq, const float* Fp, const float* radius,
                            int O, int B, uint32_t ignored, float* Dmin, const ExactOut& ex);

// What the selection (k_select, or the flush phase of k_screen) produces per horizon step
struct SelectSink {
    int* rowlist = nullptr;   // [N*O] compact list of candidate rows t*O + o
    float* listDa = nullptr;  // [N*O] their screening values (what k_exact compares its exact values with), or nullptr
    int* range = nullptr;     // [N][4] start and length of each rollout's entries in the list, tau (float bits), unused
    int* total = nullptr;     // number of listed rows (zeroed before the launch)
    int k = 0;
    float delta = 0.f;        // tau = (k-th smallest screening value) + delta
    // AUDIT sample: a non-candidate pair is recorded when (hash(pair ^ audit_seed) & audit_mask) == 0 (mask
    // 0xffffffff: never) as (row = (step_row0 + t) * O + o, screening value) in a list of the whole propagate;
    // omds_launch_audit evaluates the list in fp32 against the layer-1 slabs of all horizon steps: max (Da - D) -> maxerr_bits[2]
    unsigned audit_mask = 0xffffffffu;
    unsigned audit_seed = 0;  // changes with every horizon step and propagate, so that over time every pair gets audited
    int* audit_rows = nullptr;   // [audit_cap]
    float* audit_da = nullptr;   // [audit_cap]
    int* audit_total = nullptr;  // entries recorded so far in this propagate (may exceed audit_cap: the excess is dropped)
    int audit_cap = 0;
    int step_row0 = 0;        // (step - 1) * N: row of rollout 0 in the all-steps layer-1 table k_audit reads
};
void omds_launch_select(hipStream_t s, const float* Dmin, int B, int O, const SelectSink& sel);
void omds_launch_audit(hipStream_t s, const MlpDev& m, const float* FqAll, const float* Fp, const float* radius, int O,
                       uint32_t ignored, const int* rows, const float* da, const int* total, int cap, unsigned* maxerr_bits);
// calibration of the screening bound on the device: the batch of states, and max |x - y| into *out_bits (float bits, atomicMax)
void omds_launch_calib_states(hipStream_t s, float* qT, int B, int n, const float* lo, const float* hi, const float* center,
                              const float* trajT, int N, int H, unsigned seed);
void omds_launch_max_abs_diff(hipStream_t s, const float* x, const float* y, long long n, unsigned* out_bits);
// sweep statistics (omds.h: omds_screen_sweep_hist), word indices of the 64-bit histogram buffer
constexpr int OMDS_HIST_LOG_BINS = OMDS_SWEEP_HIST_LOG_BINS, OMDS_HIST_RATIO_BINS = OMDS_SWEEP_HIST_RATIO_BINS;
constexpr int OMDS_HIST_PAIRS = 0, OMDS_HIST_NONCAND = 1, OMDS_HIST_ABOVE_HALF = 2, OMDS_HIST_ABOVE_EPS = 3, OMDS_HIST_NONFINITE = 4,
              OMDS_HIST_MAX_POS = 5, OMDS_HIST_MAX_ABS = 6, OMDS_HIST_STEPS = 7, OMDS_HIST_BINS0 = 8;
static_assert(OMDS_HIST_BINS0 + 2 * OMDS_HIST_LOG_BINS + OMDS_HIST_RATIO_BINS == OMDS_SWEEP_HIST_WORDS, "omds.h: OMDS_SWEEP_HIST_WORDS");
void omds_launch_sweep_hist(hipStream_t s, const float* D, const float* Da, const int* range, int N, int O, float eps,
                            unsigned long long* hist, unsigned* maxabs_bits);
void omds_launch_exact(hipStream_t s, const MlpDev& m, const float* Fq, const float* Fp, const float* radius, int O,
                       int B, uint32_t ignored, float* Dmin, const int* rowlist, const int* total, unsigned* maxerr_bits,
                       const ExactOut& ex);

// ---- launchers implemented in rollout_kernels.hip ---------------------------------------------
void omds_launch_modulate(hipStream_t s, const StepArgs& a);
// fused per-step tail (tail_kernel.hip): top-k + pass 2 + blend + modulation + next-step layer-1 half
bool omds_tail_supported(int n_dof, int k);
int omds_tail_scratch_rows(int N, int k);
int omds_tail_rows(int N, int k, bool g4_ok = false);   // pass-2 tile height (16 | 32; 4 = 4-row groups, only with g4_ok) for N rollouts with k closest obstacles   // rows of the tanh-derivative scratch the tail may touch (either tile height)
// FqOut: where the next step's encoded joint inputs go (nullptr: in place).  guard_range / e_bound / viol: screened tanh step --
// Dmin holds exact values on the candidates and screening values elsewhere; the tail counts the rollouts whose k-th smallest
// value is not e_bound below k_select's tau (range[4 t + 2]) into *viol
void omds_launch_tail(hipStream_t s, const MlpDev& m, const float* Fp, const float* radius, const float* xyzr,
                      const float* Dmin, float* Fq, float* dscr, int O, const StepArgs& st, int t_begin, int t_end,
                      uint16_t* FqH = nullptr, int ldF = 0, float* FqOut = nullptr, const int* guard_range = nullptr,
                      float e_bound = 0.f, unsigned* viol = nullptr);
// screened step's tail: top-k over the candidates k_exact evaluated + pass-2 backward on its masks + the rest of k_tail
bool omds_tail_sel_supported(int n_dof, int k);
int omds_cu_count();   // CUs of the current device (asked once per device)
#ifdef OMDS_TEST_HOOKS
void omds_force_tile_rows(int tail_sel_rows, int tail_rows);   // test hook (libomds_hip_test.so): 0 = the launcher's own choice
#endif
void omds_launch_tail_sel(hipStream_t s, const MlpDev& m, const float* Fp, const float* radius, const float* xyzr,
                          float* Fq, int O, const StepArgs& st, const int* rowlist, const int* range, const ExactOut& ex,
                          uint16_t* FqH, int ldF, float e_bound, unsigned* viol);
// fused one-launch step for scenes with few obstacles (step_small.hip): rollouts per workgroup, 0 = scene does not qualify
int omds_step_small_rollouts(const MlpDev& m, int n_dof, int O, int k);
void omds_launch_step_small(hipStream_t s, const MlpDev& m, const float* Fp, const float* radius, const float* xyzr, float* Fq,
                            int O, uint32_t ignored, const StepArgs& st);
void omds_launch_net_small(hipStream_t s, const MlpDev& m, const float* Fp, const float* radius, const float* xyzr, float* Fq,
                           int O, uint32_t ignored, int n_dof, int k, const float* qT, int ldq, int B, float* gradx, float* drow,
                           int32_t* idx, float* Dmin);
struct CostArgs {
    int N, H, n;
    uint32_t terms;   // OMDS_COST_* bits
    const float* trajT; const float* distT; float* cost;
    float qf[OMDS_MAX_DOF], qmin[OMDS_MAX_DOF], qmax[OMDS_MAX_DOF];
    float dh[(OMDS_MAX_DOF + 1) * 4];
    float goal_fk[OMDS_MAX_DOF * 3];
};
void omds_launch_cost(hipStream_t s, const CostArgs& a);
void omds_host_link_endpoints(const float* q, const float* dh, int n, float* pts);
// reductions: red layout documented in rollout_kernels.hip
void omds_launch_cost_sum(hipStream_t s, const float* cost, int N, float* red2);
void omds_launch_weights(hipStream_t s, const float* cost, int N, const float* red2_global, float* w, float* red_sumw);
int omds_red_size(int K, int n);
void omds_launch_policy_sums(hipStream_t s, int N, int n, int K, const float* w, const float* muT, const float* sigmaT,
                             const float* alphaT, const float* maxact, const float* phisum0, const float* qdotT,
                             const float* cost, int include_rollout0, float* red);
void omds_launch_sample(hipStream_t s, int N, int n, int K, const float* means, float mu_s, float sigma_s, float alpha_s,
                        uint64_t seed, int64_t rollout_offset, float* muT, float* sigmaT, float* alphaT);
void omds_launch_candidates(hipStream_t s, int N, int H, int n, int K, const float* trajT, const float* distT,
                            const float* dotT, const float* means, float thr_dist, float thr_kernel, float thr_dot,
                            float rbf_p, unsigned char* flags, int* counts, int* offsets, int cap, float* cand_q,
                            int* cand_th);
void omds_launch_broadcast_q(hipStream_t s, const float* q_host_vals, int n, int N, float* dstT);
// layout conversions between reference (AoS) and device (SoA) orders
void omds_launch_transpose(hipStream_t s, const float* src, float* dst, int rows, int cols);  // dst[c][r] = src[r][c]
void omds_launch_permute_hxn_to_nhx(hipStream_t s, const float* srcT, float* dst, int H, int X, int N, int Xld);
void omds_launch_update_inputs(hipStream_t s, const float* kval, const float* act, int N, int H, int K, int kval_is_product, float* maxact,
                               float* phisum0);
void omds_launch_gather_rows(hipStream_t s, const float* srcT, float* dst, const int* tlist, int count, int H, int X, int N, int Xld);

