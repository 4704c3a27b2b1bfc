/* Minimal pure-C client of the C ABI (include/mcl.h): what a roscpp / C host links against.
 *   gcc -std=c11 -Iinclude examples/mcl_c_example.c -Lsmarc_navigation_amd -lmcl_hip \
 *       -Wl,-rpath,$PWD/smarc_navigation_amd -lm -o /tmp/mcl_c_example && /tmp/mcl_c_example
 * Prints one line of numbers that tests/test_gpu_c_client.py compares with the ctypes path. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "mcl.h"

#define CHECK(call)                                                              \
  do {                                                                           \
    int rc_ = (call);                                                            \
    if (rc_ != MCL_OK) {                                                         \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mcl_last_error(h));          \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

int main(void) {
  mcl_handle* h = NULL;
  mcl_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.n_particles = 4096;
  cfg.seed = 42;
  cfg.rng_mode = MCL_RNG_NATIVE;
  cfg.resample_scheme = MCL_RESAMPLE_SYSTEMATIC;
  cfg.meas_std = 1.5;
  const double init_cov[6] = {0.5, 0.5, 0, 0, 0, 0.01}, proc_cov[6] = {1e-4, 1e-4, 0, 0, 0, 1e-6};
  const double res_cov[6] = {0.01, 0.01, 0, 0, 0, 1e-5};
  memcpy(cfg.init_cov, init_cov, sizeof init_cov);
  memcpy(cfg.process_cov, proc_cov, sizeof proc_cov);
  memcpy(cfg.resample_cov, res_cov, sizeof res_cov);
  const double t[3] = {2.0, -1.0, 0.0}, q[4] = {0.0, 0.0, sin(0.15), cos(0.15)};
  if (mcl_matrix_from_tf(t, q, cfg.m2o) != MCL_OK) return 1;
  int rc = mcl_create(&cfg, &h);
  if (rc != MCL_OK) {
    fprintf(stderr, "mcl_create -> %d: %s\n", rc, mcl_last_error(NULL));
    return 2;
  }
  CHECK(mcl_init_particles(h, NULL));
  mcl_odom od;
  memset(&od, 0, sizeof od);
  od.v[0] = 1.0;
  od.v[1] = 0.05;
  od.w_z = 0.02;
  od.q[3] = 1.0;
  od.z = -2.0;
  for (int k = 0; k < 25; ++k) {
    od.stamp = 100.0 + 0.02 * (k + 1);
    CHECK(mcl_predict(h, &od, 0.02, NULL));
  }
  CHECK(mcl_update_gps(h, 2.6, -0.8));
  CHECK(mcl_resample(h, NULL, 0, NULL));
  double mean[6], yaw, cov[9];
  CHECK(mcl_mean_cov(h, mean, &yaw, cov));
  printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", mean[0], mean[1], mean[2], yaw, cov[0], cov[1], cov[4]);
  CHECK(mcl_destroy(h));
  return 0;
}
