/* grip_render.c -- CPU ORACLE (test infrastructure): placeholder, filled in below. */
#include "grip_oracle_int.h"
void orc_render(const OrcModel *m, const OrcData *d, int width, int height, unsigned char *rgb, float *depth) { (void)m; (void)d; (void)width; (void)height; (void)rgb; (void)depth; }
void orc_transform_depth(float *depth, int n, unsigned char *out) { (void)depth; (void)n; (void)out; }
void orc_observation(const OrcModel *m, const OrcEnvConfig *c, const OrcData *d, unsigned char *obs) { (void)m; (void)c; (void)d; (void)obs; }
