/*
 * oracle/csrc/fsrl_oracle.c -- CPU restatement of the float64 scalar pieces of the FSRL
 * policy-update path.  TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg as the checker; the product (fsrl_amd/) never links it.
 *
 * Each function cites the reference lines whose arithmetic it restates.  Build:
 *   gcc -O2 -ffp-contract=off -shared -fPIC -o oracle/_build/libfsrl_oracle.so fsrl_oracle.c
 * (-ffp-contract=off: the reference's numba/numpy code rounds the multiply and the add
 *  separately; a fused multiply-add would change the last bit.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* GAE(lambda) reverse scan -- fsrl/policy/base_policy.py:524-540 (gae_return).
 *   delta_i = rew_i + v_next_i*gamma - v_i          (float32 inputs promoted to float64,
 *   disc_i  = (1 - end_i) * (gamma*lambda)           numba typing: f32 array * f64 scalar)
 *   gae_i   = delta_i + disc_i * gae_{i+1}, gae_n = 0
 * v_next must already carry the ~terminated mask (base_policy.py:428). */
void fsrl_oracle_gae_return(const float* v, const float* v_next, const double* rew,
                            const uint8_t* end_flag, int64_t n, double gamma,
                            double gae_lambda, double* adv_out) {
    const double gl = gamma * gae_lambda;
    double gae = 0.0;
    for (int64_t i = n - 1; i >= 0; --i) {
        double delta = rew[i] + (double)v_next[i] * gamma - (double)v[i];
        double disc = (1.0 - (double)(end_flag[i] != 0)) * gl;
        gae = delta + disc * gae;
        adv_out[i] = gae;
    }
}

/* n-step return -- fsrl/policy/base_policy.py:543-567 (nstep_return).
 * indices is [n_step][bsz]; target_q is [bsz][q] float64 (already value-masked);
 * metric/end_flag are whole-buffer arrays.  Output overwrites target_q semantics into out. */
void fsrl_oracle_nstep_return(const double* metric, const uint8_t* end_flag,
                              const double* target_q, const int64_t* indices, int64_t bsz,
                              int64_t q, double gamma, int64_t n_step, double* out) {
    double* gamma_buffer = (double*)malloc(sizeof(double) * (size_t)(n_step + 1));
    gamma_buffer[0] = 1.0;
    for (int64_t i = 1; i <= n_step; ++i) gamma_buffer[i] = gamma_buffer[i - 1] * gamma;
    for (int64_t b = 0; b < bsz; ++b) {
        int64_t gammas = n_step;
        double ret = 0.0; /* identical across the q columns: metric is per-row */
        for (int64_t n = n_step - 1; n >= 0; --n) {
            int64_t now = indices[n * bsz + b];
            if (end_flag[now]) {
                gammas = n + 1;
                ret = 0.0;
            }
            ret = metric[now] + gamma * ret;
        }
        for (int64_t c = 0; c < q; ++c)
            out[b * q + c] = target_q[b * q + c] * gamma_buffer[gammas] + ret;
    }
    free(gamma_buffer);
}

/* PID Lagrange multiplier -- fsrl/utils/optim_util.py:28-41 (LagrangianOptimizer.step).
 * state = {error_old, error_integral, lagrangian}; returns the new multiplier. */
double fsrl_oracle_pid_step(double state[3], const double pid[3], double value,
                            double threshold) {
    double error_new = value - threshold;
    double diff = error_new - state[0];
    double error_diff = diff > 0.0 ? diff : 0.0;
    double integ = state[1] + error_new;
    state[1] = integ > 0.0 ? integ : 0.0;
    state[0] = error_new;
    double lag = pid[0] * error_new + pid[1] * state[1] + pid[2] * error_diff;
    state[2] = lag > 0.0 ? lag : 0.0;
    return state[2];
}
