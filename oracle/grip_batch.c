/* grip_batch.c -- CPU ORACLE (test infrastructure): OpenMP loop over independent envs, used only
 * by bench.py's cpu_baseline leg and by tests. */
#include "grip_oracle_int.h"
#include <omp.h>

int orc_num_threads(void) { return omp_get_max_threads(); }

/* steps n envs once each; actions [n][6]; obs (may be NULL) [n][5*64*64]; returns total mj substeps */
long orc_batch_env_step(const OrcModel *m, const OrcEnvConfig *c, OrcEnv *envs, int n, const double *actions,
                        OrcStepOut *outs, int auto_reset, int threads, unsigned char *obs) {
    long total = 0;
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (int i = 0; i < n; i++) {
        orc_env_step(m, c, envs + i, actions + 6 * i, outs + i);
        total += outs[i].n_substeps;
        if (auto_reset && outs[i].done) { OrcStepOut tmp; orc_env_reset(m, c, envs + i, &tmp); }
        if (obs) orc_observation(m, c, &envs[i].d, obs + (size_t)i * 5 * 64 * 64);
    }
    return total;
}
