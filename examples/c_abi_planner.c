/*
 * One planner iteration of the MI355X path from plain C, through nothing but include/omds.h -- the drop-in boundary is a
 * C ABI: no Python, no torch, no C++ types.  The sequence is the reference's slow loop (ds_mppi/frankaPlanner.py:132-145):
 * sample_policy -> propagate -> get_cost -> shift_policy_means, plus get_qdot.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_abi_planner.c -o c_abi_planner -Loptimalmodulationds_amd/csrc -lomds_hip \
 *       -Wl,-rpath,$PWD/optimalmodulationds_amd/csrc -lm
 *   ./c_abi_planner model.bin N H        (model.bin: written by tests/test_gpu_c_abi.py -- a flat export of the network, the scene and a policy)
 *
 * model.bin layout (little-endian): int32 n_dof, n_linear, dims[n_linear + 1], n_obs, K; then float32 W_0 [dims1 x dims0], b_0, ...,
 * obstacles [n_obs x 4], q_cur [n], q_goal [n], dh_params [(n + 1) x 4], q_min [n], q_max [n], mu_c [K x n], sigma_c [K], alpha_c [K x n].
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "omds.h"

#define CHECK(call)                                                                    \
    do {                                                                               \
        int rc_ = (call);                                                              \
        if (rc_ != OMDS_OK) {                                                          \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, omds_last_error(ctx)); \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

static float* read_floats(FILE* f, size_t n) {
    float* p = (float*)malloc((n ? n : 1) * sizeof(float));
    if (!p || fread(p, sizeof(float), n, f) != n) { fprintf(stderr, "model file too short\n"); exit(2); }
    return p;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s model.bin n_rollouts horizon\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    const int N = atoi(argv[2]), H = atoi(argv[3]);
    int32_t n_dof, n_linear, dims[16], n_obs, K;
    if (fread(&n_dof, 4, 1, f) != 1 || fread(&n_linear, 4, 1, f) != 1 || n_linear < 2 || n_linear > 15 ||
        fread(dims, 4, (size_t)n_linear + 1, f) != (size_t)n_linear + 1 || fread(&n_obs, 4, 1, f) != 1 || fread(&K, 4, 1, f) != 1) {
        fprintf(stderr, "bad model header\n");
        return 2;
    }
    const float* W[16];
    const float* b[16];
    for (int i = 0; i < n_linear; ++i) {
        W[i] = read_floats(f, (size_t)dims[i] * dims[i + 1]);
        b[i] = read_floats(f, (size_t)dims[i + 1]);
    }
    float* obs = read_floats(f, (size_t)n_obs * 4);
    float* q_cur = read_floats(f, n_dof);
    float* q_goal = read_floats(f, n_dof);
    float* dh = read_floats(f, (size_t)(n_dof + 1) * 4);
    float* q_min = read_floats(f, n_dof);
    float* q_max = read_floats(f, n_dof);
    float* mu_c = read_floats(f, (size_t)K * n_dof);
    float* sigma_c = read_floats(f, K);
    float* alpha_c = read_floats(f, (size_t)K * n_dof);
    fclose(f);

    omds_ctx* ctx = NULL;
    omds_config cfg = {n_dof, N, H, 50, n_obs > 64 ? n_obs : 64, 5, 0, 0};   /* MPPI.__init__ (MPPI.py:22-66) */
    CHECK(omds_create(&cfg, &ctx));
    CHECK(omds_set_mlp(ctx, n_linear, dims, W, b, OMDS_ACT_RELU, dims[n_linear] == 9 ? 100.f : 1.f));   /* robot_sdf.py:31-51, MPPI.py:236 */
    CHECK(omds_set_obstacles(ctx, obs, n_obs));                                                          /* MPPI.update_obstacles */
    omds_params prm;
    omds_default_params(&prm);
    prm.dt = 0.5f;                    /* config.yaml:44-45 */
    prm.dst_thr = 0.01f;
    prm.ignored_links = 7u;           /* MPPI.py:62: links 0, 1, 2 */
    CHECK(omds_set_params(ctx, &prm));
    CHECK(omds_set_ds(ctx, q_goal));                                        /* LinDS(q_f) */
    CHECK(omds_set_cost(ctx, dh, q_min, q_max));                            /* Cost(q_f, dh_params) */

    CHECK(omds_sample_policy(ctx, mu_c, sigma_c, alpha_c, 0.f, 0.f, 3.f, K, 4242u, 0));   /* Policy.sample_policy */
    CHECK(omds_propagate(ctx, q_cur, 0));                                                  /* MPPI.propagate */
    float* cost = (float*)malloc((size_t)N * sizeof(float));
    CHECK(omds_cost(ctx, cost));                                                           /* MPPI.get_cost */
    int32_t mask[50];
    CHECK(omds_weighted_update(ctx, 0.1f, 0.1f, mu_c, sigma_c, alpha_c, mask, NULL));      /* MPPI.shift_policy_means */
    float qdot[OMDS_MAX_DOF];
    CHECK(omds_get_qdot(ctx, 1, qdot));                                                    /* MPPI.get_qdot('weighted') */

    double csum = 0.0;
    int upd = 0;
    for (int t = 0; t < N; ++t) csum += cost[t];
    for (int k = 0; k < K; ++k) upd += mask[k];
    printf("version %d\ncost_sum %.9g\nupdated %d\nqdot", omds_version(), csum, upd);
    for (int j = 0; j < n_dof; ++j) printf(" %.9g", qdot[j]);
    printf("\nmu0");
    for (int j = 0; j < n_dof && K > 0; ++j) printf(" %.9g", mu_c[j]);
    printf("\n");
    omds_destroy(ctx);
    return 0;
}
