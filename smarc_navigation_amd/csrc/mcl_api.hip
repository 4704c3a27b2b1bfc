// mcl_api.hip -- host side of libmcl_hip.so: the C ABI declared in include/mcl.h.
// C++ host code that owns the device buffers, orders kernels on one HIP stream per handle and
// runs the shard-exchange steps over RCCL (one process per GPU) or device copies (LOCAL group).
// The machinery is in mcl_host*.h (one translation unit); this file is the entry points.
#include "mcl_host.h"
#include "mcl_host_resample.h"
#include "mcl_host_moments.h"
#include "mcl_host_update.h"


// dead-reckoning integrator (host only; uses euler_from_quat above)
#include "mcl_dr_impl.h"
// bathymetry map builder (uses rot_rpy above)
#include "mcl_gridmap.h"

// ============================================================================================ C ABI
extern "C" {

int mcl_abi_version(void) { return MCL_ABI_VERSION; }

const char* mcl_status_string(int s) {
  switch (s) {
    case MCL_OK: return "ok";
    case MCL_ERR_INVALID: return "invalid argument";
    case MCL_ERR_NO_DEVICE: return "no gfx950 HIP device";
    case MCL_ERR_HIP: return "HIP runtime error";
    case MCL_ERR_UNSUPPORTED: return "unsupported";
    case MCL_ERR_STATE: return "bad call order";
    case MCL_ERR_COMM: return "RCCL error";
    case MCL_ERR_ALLOC: return "allocation failed";
  }
  return "unknown status";
}

const char* mcl_last_error(const mcl_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int mcl_matrix_from_tf(const double translation[3], const double quaternion[4], double m16[16]) {
  return matrix_from_tf_impl(translation, quaternion, m16);
}

int mcl_device_count(int* count) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  if (count) *count = n;
  return MCL_OK;
}

int mcl_create(const mcl_config* cfg, mcl_handle** out) {
  if (!cfg || !out) {
    g_create_err = "mcl_create: null argument";
    return MCL_ERR_INVALID;
  }
  *out = nullptr;
  if (cfg->n_particles < 1 || cfg->n_particles > 0x7fffffffll) {
    g_create_err = "mcl_create: n_particles out of range";
    return MCL_ERR_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    g_create_err = "mcl_create: no HIP device visible (this library has no CPU fallback)";
    return MCL_ERR_NO_DEVICE;
  }
  if (cfg->device < 0 || cfg->device >= ndev) {
    g_create_err = "mcl_create: device ordinal out of range";
    return MCL_ERR_INVALID;
  }
  hipDeviceProp_t prop;
  memset(&prop, 0, sizeof prop);
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_err = std::string("mcl_create: device is not gfx950 (found ") + prop.gcnArchName + ")";
    return MCL_ERR_NO_DEVICE;
  }
  mcl_handle* h = new mcl_handle();
  h->cfg = *cfg;
  h->n = cfg->n_particles;
  h->world = cfg->world > 1 ? cfg->world : 1;
  h->rank = h->world > 1 ? cfg->rank : 0;
  h->ng = cfg->n_global > 0 ? cfg->n_global : cfg->n_particles;
  h->goff = h->world > 1 ? cfg->global_offset : 0;
  h->device = cfg->device;
  memset(&h->tacc, 0, sizeof h->tacc);
  {
    auto on = [](const char* name) {
      const char* v = getenv(name);
      return v && v[0] == '1';
    };
    h->env_debug_work = getenv("MCL_DEBUG_WORK") != nullptr;
    if (const char* sv = getenv("MCL_SORT_VISITS")) h->env_sort = sv[0] == '1' ? 1 : 0;
    if (const char* sv = getenv("MCL_SWEEP")) h->env_sweep = sv[0] == '1' ? 1 : 0;
    if (const char* sv = getenv("MCL_SLICE")) h->env_slice = sv[0] == '1' ? 1 : 0;
    if (const char* sv = getenv("MCL_SLICE_GROUP")) h->env_slice_group = sv[0] == '1' ? 1 : 0;
    if (const char* sv = getenv("MCL_HANDOVER_SLICE")) h->env_handover_slice = sv[0] == '1' ? 1 : 0;
    if (const char* sv = getenv("MCL_VISIT")) h->env_visit = sv[0] == '1' ? 1 : 0;
    if (const char* sv = getenv("MCL_VISIT_BINS")) {
      int b[3] = {0, 0, 0};
      if (sscanf(sv, "%d,%d,%d", &b[0], &b[1], &b[2]) == 3 && b[0] >= 1 && b[1] >= 1 && b[2] >= 1 &&
          (long long)b[0] * b[1] * b[2] <= VISIT_MAX_BINS && (b[0] * b[1] * b[2]) % 64 == 0)
        for (int c = 0; c < 3; ++c) h->visit_nb[c] = b[c];
    }
    if (const char* sv = getenv("MCL_VISIT_RANGE")) {
      const double r = atof(sv);
      if (r >= 0.5 && r <= 16.0) h->visit_range = (float)r;
    }
    if (const char* sv = getenv("MCL_VISIT_MIN_N")) h->visit_min_n = std::max(1ll, atoll(sv));
    if (const char* sv = getenv("MCL_SWEEP_NSUB")) h->env_nsub = (sv[0] == '2' || sv[0] == '4') ? sv[0] - '0' : 1;
    h->env_force_comm = on("MCL_FORCE_COMM");
    if (const char* ex = getenv("MCL_EXCHANGE")) h->exch_allgather = strcmp(ex, "allgather") == 0;
    if (const char* fi = getenv("MCL_FAULT_INJECT")) h->fault_step = strcmp(fi, "step_after_predict") == 0;
    h->env_no_overlap = on("MCL_NO_OVERLAP");
  }
  if (h->ng > 0xffffffffll || h->goff + h->n > h->ng || h->rank >= h->world) {
    g_create_err = "mcl_create: inconsistent shard geometry";
    delete h;
    return MCL_ERR_INVALID;
  }
  if (h->world > 1 && (h->ng != h->n * h->world || h->goff != h->n * h->rank)) {
    g_create_err = "mcl_create: shards must be equal-sized contiguous blocks (n_global = world * n_particles)";
    delete h;
    return MCL_ERR_INVALID;
  }
#define CREATE_CHK(call)                                                   \
  do {                                                                     \
    hipError_t e_ = (call);                                                \
    if (e_ != hipSuccess) {                                                \
      g_create_err = std::string(#call " failed: ") + hipGetErrorString(e_); \
      mcl_destroy(h);                                                      \
      return e_ == hipErrorOutOfMemory ? MCL_ERR_ALLOC : MCL_ERR_HIP;      \
    }                                                                      \
  } while (0)
  CREATE_CHK(hipSetDevice(h->device));
  CREATE_CHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  const size_t n = (size_t)h->n, ng = (size_t)h->ng;
  CREATE_CHK(hipMalloc(&h->state[0], sizeof(double) * 6 * n));
  CREATE_CHK(hipMalloc(&h->state[1], sizeof(double) * 6 * n));
  CREATE_CHK(hipMemsetAsync(h->state[0], 0, sizeof(double) * 6 * n, h->stream));
  CREATE_CHK(hipMemsetAsync(h->state[1], 0, sizeof(double) * 6 * n, h->stream));
  if (h->world > 1 && h->exch_allgather) CREATE_CHK(hipMalloc(&h->state_glob, sizeof(double) * 6 * ng));
  CREATE_CHK(hipMalloc(&h->lw, sizeof(double) * n));
  CREATE_CHK(hipMalloc(&h->q, sizeof(u64) * n));
  CREATE_CHK(hipMalloc(&h->ncum, sizeof(u32) * ng));
  CREATE_CHK(hipMalloc(&h->zcum, sizeof(u32) * ng));
  h->ntiles_loc = (h->n + MCL_SCAN_TILE - 1) / MCL_SCAN_TILE;
  h->ntiles_glob = (h->ng + MCL_SCAN_TILE - 1) / MCL_SCAN_TILE;
  CREATE_CHK(hipMalloc(&h->tile64, sizeof(u64) * (size_t)(h->ntiles_loc + 1)));
  CREATE_CHK(hipMalloc(&h->tile32, sizeof(u32) * (size_t)(h->ntiles_glob + 1)));
  CREATE_CHK(hipMalloc(&h->part, sizeof(double) * MOM_COUNT * MCL_MAX_GRID));
  CREATE_CHK(hipMalloc(&h->scal, sizeof(double) * 64));
  CREATE_CHK(hipMemsetAsync(h->scal, 0, sizeof(double) * 64, h->stream));
  CREATE_CHK(hipMalloc(&h->zr, sizeof(u32) * n));
  CREATE_CHK(hipMalloc(&h->dupes32, sizeof(u32) * ng));
  CREATE_CHK(hipMalloc(&h->desc, sizeof(u64) * (size_t)(h->ntiles_glob + 1)));
  CREATE_CHK(hipMemsetAsync(h->desc, 0, sizeof(u64) * (size_t)(h->ntiles_glob + 1), h->stream));
  CREATE_CHK(hipMalloc(&h->ctrl, CTRL_BYTES));
  CREATE_CHK(hipMemsetAsync(h->ctrl, 0, CTRL_BYTES, h->stream));
  CREATE_CHK(hipMalloc(&h->totals, sizeof(u64) * (size_t)(h->world + 1)));
  CREATE_CHK(hipMemsetAsync(h->totals, 0, sizeof(u64) * (size_t)(h->world + 1), h->stream));
  CREATE_CHK(hipHostMalloc(&h->host_pin, sizeof(double) * RING_STRIDE * MEAN_RING, hipHostMallocDefault));
  if (hipHostGetDevicePointer((void**)&h->host_pin_dev, h->host_pin, 0) != hipSuccess) h->host_pin_dev = nullptr;
  CREATE_CHK(hipStreamSynchronize(h->stream));
#undef CREATE_CHK
  *out = h;
  return MCL_OK;
}

int mcl_destroy(mcl_handle* h) {
  if (!h) return MCL_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  t_collect(h);
  for (auto& e : h->ev_pool) {
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  if (h->comm2) ncclCommDestroy(h->comm2);
  if (h->comm) ncclCommDestroy(h->comm);
  h->comm2 = h->comm = nullptr;
  if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
  if (h->ev_state_ready) (void)hipEventDestroy(h->ev_state_ready);
  if (h->ev_gather_done) (void)hipEventDestroy(h->ev_gather_done);
  void* bufs[] = {h->state[0], h->state[1], h->state_glob, h->lw, h->wnorm, h->q, h->ncum, h->zcum, h->zr, h->dupes32, h->desc, h->ctrl,
                  h->tile64, h->tile32, h->part, h->scal, h->totals, h->idx, h->replay_dev, h->pose7,
                  h->beam_sc, h->ranges_dev, h->exp_dev, h->grid, h->pose_dev, h->mbes_worklist, h->mbes_groups, h->sort_keys, h->sort_keys_out, h->sort_idx, h->mbes_perm, h->sort_tmp, h->sweep_buf[0], h->sweep_buf[1], h->defer_idx, h->defer2_idx, h->visit_okey, h->visit_base, h->visit_cnt, h->visit_desc, h->visit_par, h->slice_loose, h->reasons_dev, h->grid_pad, h->lm_worklist, h->cq, h->u53, h->cnt, h->first,
                  h->flags, h->fcum, h->copies, h->ccum, h->dupes, h->cs, h->chunk, h->uni_dev, h->lsx, h->xsend, h->xrecv, h->shrec, h->tile_bits};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (h->mesh) mesh_free(h->mesh);
  if (h->landmarks) landmarks_free(h->landmarks);
  if (h->det_dev) (void)hipFree(h->det_dev);
  if (h->host_pin) (void)hipHostFree(h->host_pin);
  if (h->lsx_host) (void)hipHostFree(h->lsx_host);
  if (h->work_host) (void)hipHostFree(h->work_host);
  if (h->copy_stream) {
    (void)hipStreamSynchronize(h->copy_stream);
    (void)hipStreamDestroy(h->copy_stream);
  }
  for (int k = 0; k < 2; ++k) {
    if (h->sweep_stage[k]) (void)hipHostFree(h->sweep_stage[k]);
    if (h->ev_stage[k]) (void)hipEventDestroy(h->ev_stage[k]);
  }
  for (auto& e : h->ev_upd)
    if (e) (void)hipEventDestroy(e);
  for (auto& sl : h->pin_ring) {
    if (sl.ev) (void)hipEventDestroy(sl.ev);
    if (sl.p) (void)hipHostFree(sl.p);
  }
  if (h->asg_dev) (void)hipFree(h->asg_dev);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return MCL_OK;
}

int mcl_init_particles(mcl_handle* h, const double* replay_normals) {
  if (!h) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  const double* rp = nullptr;
  if (h->cfg.rng_mode == MCL_RNG_REPLAY) {
    if (!replay_normals) return fail(h, MCL_ERR_INVALID, "init_particles: REPLAY mode needs n x 6 normals");
    RET_IF(upload_replay(h, replay_normals));
    rp = h->replay_dev;
  }
  RET_IF(cancel_state_gather(h));
  h->uni_valid = false;
  h->visit_ready = false;
  NoiseArgs a = noise_args(h, h->cfg.init_cov, 0u, 0u);
  t_begin(h, MCL_K_NOISE);
  k_add_noise<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, a, rp, 1);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  h->step_predict = 0;
  h->step_resample = 0;
  h->have_lw = h->have_cdf = false;
  return MCL_OK;
}

int mcl_predict(mcl_handle* h, const mcl_odom* odom, double dt, const double* replay_normals) {
  if (!h || !odom) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  RET_IF(cancel_state_gather(h));
  return do_predict(h, odom, dt, replay_normals);
}

int mcl_update_gps(mcl_handle* h, double gx_map, double gy_map) {
  if (!h) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  if (!(h->cfg.meas_std > 0.0)) return fail(h, MCL_ERR_INVALID, "update_gps: meas_std must be > 0");
  GpsArgs a;
  for (int k = 0; k < 4; ++k) {
    a.r0[k] = h->cfg.m2o[k];
    a.r1[k] = h->cfg.m2o[4 + k];
  }
  const double s2 = h->cfg.meas_std * h->cfg.meas_std;
  a.gx = gx_map;
  a.gy = gy_map;
  a.inv_s2 = 1.0 / s2;
  a.lognorm = std::log(2.0 * MCL_PI * s2);
  t_begin(h, MCL_K_UPDATE_GPS);
  k_gps_logw<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, a, h->lw);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  h->weight_mode = MCL_WEIGHT_LINEAR_FLOOR;
  h->have_lw = true;
  h->max_valid = false;
  h->residual_k = -1;
  return MCL_OK;
}

int mcl_set_map_grid(mcl_handle* h, const float* z, int32_t nx, int32_t ny, double ox, double oy, double res) {
  if (!h || !z || nx < 2 || ny < 2 || !(res > 0.0)) return fail(h, MCL_ERR_INVALID, "set_map_grid: bad argument");
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (nx > (1 << 21) || ny > (1 << 21)) return fail(h, MCL_ERR_UNSUPPORTED, "set_map_grid: more than 2^21 nodes a side");
  if (h->grid) (void)hipFree(h->grid);
  if (h->grid_pad) (void)hipFree(h->grid_pad);
  h->grid = nullptr;
  h->grid_pad = nullptr;
  const size_t cnt = (size_t)nx * (size_t)ny;
  HIPCHK(h, hipMalloc(&h->grid, sizeof(float) * cnt));
  HIPCHK(h, hipMemcpy(h->grid, z, sizeof(float) * cnt, hipMemcpyHostToDevice));
  HIPCHK(h, upload_padded_heights(z, nx, ny, &h->grid_pad));
  float mn = z[0], mx = z[0];
  for (size_t k = 1; k < cnt; ++k) {
    if (z[k] < mn) mn = z[k];
    if (z[k] > mx) mx = z[k];
  }
  h->gnx = nx;
  h->gny = ny;
  h->gox = ox;
  h->goy = oy;
  h->gres = res;
  h->gzmin = mn;
  h->gzmax = mx;
  {
    // steepest gradient of a bilinear patch: its x slope lies between those of the cell's two x edges, its y slope
    // between those of the two y edges
    double g2 = 0.0;
    for (int ix = 0; ix + 1 < nx; ++ix)
      for (int iy = 0; iy + 1 < ny; ++iy) {
        const size_t k = (size_t)ix * ny + iy;
        const double h00 = z[k], h01 = z[k + 1], h10 = z[k + ny], h11 = z[k + ny + 1];
        const double ax = std::max(std::fabs(h10 - h00), std::fabs(h11 - h01));
        const double ay = std::max(std::fabs(h01 - h00), std::fabs(h11 - h10));
        g2 = std::max(g2, ax * ax + ay * ay);
      }
    h->gslope_max = std::sqrt(g2) / res;
  }
  h->map_kind = 0;
  return MCL_OK;
}

int mcl_set_map_mesh(mcl_handle* h, const float* verts, int64_t nv, const uint32_t* tris, int64_t nt) {
  return mcl_set_map_mesh_ex(h, verts, nv, tris, nt, 0u);
}

int mcl_set_map_mesh_ex(mcl_handle* h, const float* verts, int64_t nv, const uint32_t* tris, int64_t nt,
                        uint32_t flags) {
  if (!h || !verts || !tris || nv < 3 || nt < 1) return fail(h, MCL_ERR_INVALID, "set_map_mesh: bad argument");
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->mesh) mesh_free(h->mesh);
  h->mesh = nullptr;
  std::string err;
  // (a structured mesh is cast as a soup only on request: the slice's vertex records are built then)
  int rc = mesh_build(verts, nv, tris, nt, (flags & (MCL_MESH_GENERAL | MCL_MESH_UNSTRUCTURED)) != 0, &h->mesh, &err);
  if (rc != MCL_OK) {
    h->err = err;
    return rc;
  }
  h->map_kind = 1;
  h->mesh_heightfield = (flags & MCL_MESH_HEIGHTFIELD) != 0;
  h->force_general_mesh = (flags & (MCL_MESH_GENERAL | MCL_MESH_UNSTRUCTURED)) != 0;
  h->mesh_no_sweep = (flags & MCL_MESH_GENERAL) != 0;
  if (h->mesh_heightfield && h->mesh->n_vertical > 0) {
    h->err = "set_map_mesh: MCL_MESH_HEIGHTFIELD declared but the mesh has vertical faces";
    h->mesh_heightfield = false;
    return MCL_ERR_INVALID;
  }
  return MCL_OK;
}

int mcl_update_mbes(mcl_handle* h, const float* ranges, const float* beam_angles, int32_t B, double sigma,
                    double r_max, const double sensor_offset[6]) {
  if (!h || !ranges || !beam_angles || B < 1 || !(sigma > 0.0) || !(r_max > 0.0))
    return fail(h, MCL_ERR_INVALID, "update_mbes: bad argument");
  RET_IF(set_device(h));
  RET_IF(upload_beams(h, ranges, beam_angles, B));
  RET_IF(launch_mbes(h, true, B, sigma, r_max, sensor_offset, h->lw, nullptr, 0, 0));
  h->weight_mode = MCL_WEIGHT_LOG_SHIFT;
  h->have_lw = true;
  h->residual_k = -1;
  return MCL_OK;
}

int mcl_mbes_expected(mcl_handle* h, int64_t first, int64_t count, const float* beam_angles, int32_t B,
                      double r_max, const double sensor_offset[6], float* out) {
  if (!h || !beam_angles || !out || B < 1 || first < 0 || count < 1 || first + count > h->n)
    return fail(h, MCL_ERR_INVALID, "mbes_expected: bad argument");
  RET_IF(set_device(h));
  RET_IF(upload_beams(h, nullptr, beam_angles, B));
  const size_t need = (size_t)count * (size_t)B;
  if (need > h->exp_cap) {
    if (h->exp_dev) (void)hipFree(h->exp_dev);
    h->exp_dev = nullptr;
    HIPCHK(h, hipMalloc(&h->exp_dev, sizeof(float) * need));
    h->exp_cap = need;
  }
  RET_IF(launch_mbes(h, false, B, 1.0, r_max, sensor_offset, nullptr, h->exp_dev, first, count));
  HIPCHK(h, hipMemcpyAsync(out, h->exp_dev, sizeof(float) * need, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MCL_OK;
}

int mcl_set_landmarks(mcl_handle* h, const double* xyz, int64_t n_landmarks) {
  if (!h || !xyz || n_landmarks < 1) return fail(h, MCL_ERR_INVALID, "set_landmarks: bad argument");
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->landmarks) landmarks_free(h->landmarks);
  h->landmarks = new LandmarkDev();
  h->landmarks->host_xyz.assign(xyz, xyz + 3 * n_landmarks);
  return MCL_OK;
}

namespace {
// largest eigenvalue bound of a symmetric 3x3 (xx xy xz yy yz zz): Gershgorin
double sym3_lam_bound(const double* s) {
  const double r0 = s[0] + std::fabs(s[1]) + std::fabs(s[2]), r1 = s[3] + std::fabs(s[1]) + std::fabs(s[4]),
               r2 = s[5] + std::fabs(s[2]) + std::fabs(s[4]);
  return std::max(r0, std::max(r1, r2));
}
double sym3_det(const double* s) {
  return s[0] * (s[3] * s[5] - s[4] * s[4]) - s[1] * (s[1] * s[5] - s[4] * s[2]) + s[2] * (s[1] * s[4] - s[3] * s[2]);
}
// measurement covariance of this update: the Q of mcl_set_landmark_noise, else sigma^2 I
void landmark_q(const LandmarkDev* L, double sigma, double Q[6]) {
  if (L->have_q) {
    for (int k = 0; k < 6; ++k) Q[k] = L->Q[k];
  } else {
    Q[0] = Q[3] = Q[5] = sigma * sigma;
    Q[1] = Q[2] = Q[4] = 0.0;
  }
}
// every landmark inside the gate lies within this distance of the detection: d^2 >= |nu|^2 / lambda_max(S)
double landmark_gate_radius(const LandmarkDev* L, double sigma, double gate) {
  if (!L->maha) return sigma * std::sqrt(gate);
  double Q[6];
  landmark_q(L, sigma, Q);
  return std::sqrt(gate * (L->lam_cov_max + sym3_lam_bound(Q)));
}
void landmark_noise_args(const LandmarkDev* L, double sigma, LandmarkArgs& a) {
  a.maha = L->maha ? 1 : 0;
  a.lmcov = L->lmcov;
  landmark_q(L, sigma, a.Q);
  a.logdet_q = std::log(sym3_det(a.Q));
  a.lognorm = 1.5 * std::log(2.0 * MCL_PI) + 0.5 * a.logdet_q;  // isotropic: 3/2 log 2pi + 3 log sigma
}
}  // namespace

int mcl_set_landmark_noise(mcl_handle* h, const double* cov6, const double Q6[6]) {
  if (!h) return MCL_ERR_INVALID;
  if (!h->landmarks) return fail(h, MCL_ERR_STATE, "set_landmark_noise: no feature map (call mcl_set_landmarks first)");
  RET_IF(set_device(h));
  LandmarkDev* L = h->landmarks;
  const size_t n = L->host_xyz.size() / 3;
  L->host_cov.clear();
  L->lam_cov_max = 0.0;
  if (cov6) {
    for (size_t i = 0; i < n; ++i) {
      const double* s = cov6 + 6 * i;
      if (!(s[0] >= 0.0 && s[3] >= 0.0 && s[5] >= 0.0) || !(sym3_det(s) >= 0.0))
        return fail(h, MCL_ERR_INVALID, "set_landmark_noise: landmark covariance not positive semi-definite");
      L->lam_cov_max = std::max(L->lam_cov_max, sym3_lam_bound(s));
    }
    L->host_cov.assign(cov6, cov6 + 6 * n);
  }
  L->have_q = Q6 != nullptr;
  if (Q6) {
    if (!(sym3_det(Q6) > 0.0) || !(Q6[0] > 0.0)) return fail(h, MCL_ERR_INVALID, "set_landmark_noise: Q must be positive definite");
    for (int k = 0; k < 6; ++k) L->Q[k] = Q6[k];
  }
  L->maha = cov6 != nullptr || Q6 != nullptr;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  L->built_for = -1.0;  // the covariances travel with the next grid build
  return MCL_OK;
}

namespace {
// one landmark observation (mcl_update_landmarks' arguments)
struct LandmarkObs {
  const double* det;
  int n_det;
  double sigma;
  int k;
  double gate;
  const double* so;
};
int landmarks_upload(mcl_handle* h, const LandmarkObs& o) {
  if (o.n_det > h->det_cap) {
    if (h->det_dev) (void)hipFree(h->det_dev);
    h->det_dev = nullptr;
    HIPCHK(h, hipMalloc(&h->det_dev, sizeof(double) * 3 * (size_t)o.n_det));
    h->det_cap = o.n_det;
  }
  return upload(h, h->det_dev, o.det, sizeof(double) * 3 * (size_t)o.n_det);
}
// argument checks and the cell grid for this gate radius.  (A grid rebuild drains the
// stream -- the old arrays may still be read by a kernel in flight -- so the fused step calls this BEFORE its predict.)
// ride: the fused step -- the detections are not copied here; they wait for the beam table's staged copy of the same
// step (upload_sweep_beams; landmarks_launch copies them itself if the update took a path without that table)
int landmarks_prepare(mcl_handle* h, const LandmarkObs& o, const char* who, bool ride = false) {
  if (!o.det || o.n_det < 1 || !(o.sigma > 0.0) || o.k < 1 || o.k > LM_MAX_K || !(o.gate > 0.0))
    return fail(h, MCL_ERR_INVALID, std::string(who) + ": bad argument (1 <= k <= 4)");
  if (!h->landmarks) return fail(h, MCL_ERR_STATE, std::string(who) + ": no feature map (call mcl_set_landmarks first)");
  RET_IF(set_device(h));
  const double radius = landmark_gate_radius(h->landmarks, o.sigma, o.gate);
  if (!(h->landmarks->built_for == radius && h->landmarks->lm)) {
    std::string err;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int rc = landmarks_build(h->landmarks, radius, &err);
    if (rc != MCL_OK) {
      h->err = err;
      return rc;
    }
  }
  h->det_ride = nullptr;
  h->det_ride_dev = nullptr;
  if (ride) {
    h->det_ride = o.det;
    h->det_ride_n = o.n_det;
  }
  return MCL_OK;   // (landmarks_launch copies detections that did not ride)
}
// the k-NN landmark likelihood of every particle.  fused: inside mcl_step_mbes_landmarks -- the predict kernel of the
// same call may have left z, roll, pitch unstored (uni_deferred), and the kernel leaves max lw in the second slot set
int landmarks_launch(mcl_handle* h, const LandmarkObs& o, bool accumulate, bool fused) {
  static const double zero6[6] = {0, 0, 0, 0, 0, 0};
  const double* so = o.so ? o.so : zero6;
  // the detections: where the beam table's copy of this step left them, or (an update without that table: no sweep)
  // copied now
  const double* det = h->det_ride_dev;
  if (!det) {
    RET_IF(landmarks_upload(h, o));
    det = h->det_dev;
  }
  h->det_ride = nullptr;
  h->det_ride_dev = nullptr;
  LandmarkArgs a;
  memset(&a, 0, sizeof a);
  for (int c = 0; c < 6; ++c) a.st[c] = h->state[h->cur] + (size_t)c * h->n;
  a.n = h->n;
  for (int q = 0; q < 12; ++q) a.m2o[q] = h->cfg.m2o[q];
  for (int q = 0; q < 3; ++q) a.off_t[q] = so[q];
  rot_rpy(so[3], so[4], so[5], a.off_R);
  a.det = det;
  a.n_det = o.n_det;
  a.lm = h->landmarks->lm;
  a.cell_start = h->landmarks->cell_start;
  a.nb_cell = h->landmarks->nb_cell;
  a.nb_list = h->landmarks->nb_list;
  a.gx = h->landmarks->gx;
  a.gy = h->landmarks->gy;
  a.x0 = h->landmarks->x0;
  a.y0 = h->landmarks->y0;
  a.inv_cs = 1.0 / h->landmarks->cs;
  a.inv_s2 = 1.0 / (o.sigma * o.sigma);
  a.gate = o.gate;
  landmark_noise_args(h->landmarks, o.sigma, a);
  a.k = o.k;
  a.accumulate = accumulate ? 1 : 0;
  a.lw = h->lw;
  a.uni_mask = (fused && h->uni_deferred) ? 0x1cu : 0u;
  for (int c = 0; c < 3; ++c) a.uni[c] = h->uni_val[c];
  a.max_slots = fused ? (u64*)(h->ctrl + CTRL_SLOTS2) : nullptr;   // (zeroed with the whole block by this step's predict / pose launch)
  t_begin(h, MCL_K_UPDATE_LANDMARKS);
  long long blocks = (h->n + 255) / 256;   // (a wave per 64 particles, four waves per workgroup)
  if (blocks > 16384) blocks = 16384;
  if (a.maha)
    k_landmark_update<true><<<(unsigned)blocks, 256, 0, h->stream>>>(a);
  else
    k_landmark_update<false><<<(unsigned)blocks, 256, 0, h->stream>>>(a);
  t_end(h);
  HIPCHK(h, hipGetLastError());
  if (!accumulate) h->weight_mode = MCL_WEIGHT_LOG_SHIFT;
  h->have_lw = true;
  h->max_valid = fused;
  if (fused) h->slot_set = 1;
  h->residual_k = -1;
  return MCL_OK;
}
}  // namespace

int mcl_update_landmarks(mcl_handle* h, const double* det_xyz, int32_t n_det, double sigma, int32_t k, double gate,
                         const double sensor_offset[6], int32_t accumulate) {
  if (!h) return MCL_ERR_INVALID;
  const LandmarkObs o = {det_xyz, n_det, sigma, k, gate, sensor_offset};
  RET_IF(landmarks_prepare(h, o, "update_landmarks"));
  if (accumulate && !h->have_lw) return fail(h, MCL_ERR_STATE, "update_landmarks: nothing to accumulate onto");
  return landmarks_launch(h, o, accumulate != 0, false);
}

int mcl_update_landmarks_assign(mcl_handle* h, const double* det_xyz, int32_t n_det, double sigma, int32_t k_cand,
                                double gate, double new_mh_dist, const double sensor_offset[6], int32_t accumulate,
                                int32_t* assign_out, int64_t n_keep) {
  if (!h || !det_xyz || n_det < 1 || n_det > LM_SUB || !(sigma > 0.0) || k_cand < 1 || k_cand > LA_KC || !(gate > 0.0) ||
      !(new_mh_dist >= 0.0) || n_keep < 0 || (n_keep > 0 && !assign_out))
    return fail(h, MCL_ERR_INVALID, "update_landmarks_assign: bad argument (n_det <= 16, 1 <= k_cand <= 8)");
  if (!h->landmarks) return fail(h, MCL_ERR_STATE, "update_landmarks_assign: no feature map (call mcl_set_landmarks first)");
  if (accumulate && !h->have_lw) return fail(h, MCL_ERR_STATE, "update_landmarks_assign: nothing to accumulate onto");
  RET_IF(set_device(h));
  {
    // the cell grid depends on the gate radius only: rebuild (and drain the stream first -- the old
    // arrays may still be read by a kernel in flight) only when it changes
    const double radius = landmark_gate_radius(h->landmarks, sigma, gate);
    if (!(h->landmarks->built_for == radius && h->landmarks->lm)) {
      std::string err;
      HIPCHK(h, hipStreamSynchronize(h->stream));
      int rc = landmarks_build(h->landmarks, radius, &err);
      if (rc != MCL_OK) {
        h->err = err;
        return rc;
      }
    }
  }
  if (n_det > h->det_cap) {
    if (h->det_dev) (void)hipFree(h->det_dev);
    h->det_dev = nullptr;
    HIPCHK(h, hipMalloc(&h->det_dev, sizeof(double) * 3 * (size_t)n_det));
    h->det_cap = n_det;
  }
  RET_IF(upload(h, h->det_dev, det_xyz, sizeof(double) * 3 * (size_t)n_det));
  if (n_keep > h->n) n_keep = h->n;
  int* asg_dev = nullptr;
  if (n_keep > 0) {
    const size_t need = (size_t)n_keep * (size_t)n_det;
    if (need > h->asg_cap) {
      if (h->asg_dev) (void)hipFree(h->asg_dev);
      h->asg_dev = nullptr;
      h->asg_cap = 0;
      HIPCHK(h, hipMalloc(&h->asg_dev, sizeof(int) * need));
      h->asg_cap = need;
    }
    asg_dev = h->asg_dev;
  }
  static const double zero6[6] = {0, 0, 0, 0, 0, 0};
  const double* so = sensor_offset ? sensor_offset : zero6;
  LandmarkAssignArgs aa;
  memset(&aa, 0, sizeof aa);   // (uni_mask = 0, max_slots = nullptr: the state is read as stored)
  LandmarkArgs& a = aa.base;
  for (int c = 0; c < 6; ++c) a.st[c] = h->state[h->cur] + (size_t)c * h->n;
  a.n = h->n;
  for (int q = 0; q < 12; ++q) a.m2o[q] = h->cfg.m2o[q];
  for (int q = 0; q < 3; ++q) a.off_t[q] = so[q];
  rot_rpy(so[3], so[4], so[5], a.off_R);
  a.det = h->det_dev;
  a.n_det = n_det;
  a.lm = h->landmarks->lm;
  a.cell_start = h->landmarks->cell_start;
  a.nb_cell = h->landmarks->nb_cell;
  a.nb_list = h->landmarks->nb_list;
  a.gx = h->landmarks->gx;
  a.gy = h->landmarks->gy;
  a.x0 = h->landmarks->x0;
  a.y0 = h->landmarks->y0;
  a.inv_cs = 1.0 / h->landmarks->cs;
  a.inv_s2 = 1.0 / (sigma * sigma);
  a.gate = gate;
  landmark_noise_args(h->landmarks, sigma, a);
  a.k = k_cand;
  a.accumulate = accumulate ? 1 : 0;
  a.lw = h->lw;
  aa.orig = h->landmarks->orig;
  aa.new_mh = new_mh_dist;
  aa.k_cand = k_cand;
  aa.assign_out = asg_dev;
  aa.n_keep = n_keep;
  hipError_t le = hipSuccess;
  if (!h->lm_worklist) le = hipMalloc(&h->lm_worklist, sizeof(int) * ((size_t)h->n + 1));
  if (le == hipSuccess) {
    aa.worklist = h->lm_worklist;
    aa.work_count = h->lm_worklist + h->n;
    le = hipMemsetAsync(aa.work_count, 0, sizeof(int), h->stream);
  }
  if (le == hipSuccess) {
    t_begin(h, MCL_K_UPDATE_LANDMARKS);
    long long blocks = (h->n + LA_PER_BLOCK - 1) / LA_PER_BLOCK;
    if (blocks > 32768) blocks = 32768;
    // every particle: conflict-free answer or worklist entry; then the solver over the worklist (its grid
    // strides over the device-side count)
    k_landmark_assign<false><<<(unsigned)blocks, LA_PER_BLOCK * LM_SUB, 0, h->stream>>>(aa);
    k_landmark_assign<true><<<(unsigned)std::min<long long>(blocks, 2048), LA_PER_BLOCK * LM_SUB, 0, h->stream>>>(aa);
    t_end(h);
    le = hipGetLastError();
  }
  if (le == hipSuccess && n_keep > 0)
    le = hipMemcpyAsync(assign_out, asg_dev, sizeof(int) * (size_t)n_keep * n_det, hipMemcpyDeviceToHost, h->stream);
  if (le == hipSuccess && n_keep > 0) le = hipStreamSynchronize(h->stream);
  HIPCHK(h, le);
  if (!accumulate) h->weight_mode = MCL_WEIGHT_LOG_SHIFT;
  h->have_lw = true;
  h->max_valid = false;
  h->residual_k = -1;
  return MCL_OK;
}

int mcl_resample(mcl_handle* h, const double* uniforms, int64_t n_uniforms, const double* replay_normals) {
  if (!h) return MCL_ERR_INVALID;
  if (h->world > 1 && !h->comm)
    return fail(h, MCL_ERR_STATE, "resample: multi-shard handle needs mcl_comm_init or mcl_group_resample");
  if (h->cfg.rng_mode == MCL_RNG_REPLAY && !replay_normals)
    return fail(h, MCL_ERR_INVALID, "resample: REPLAY mode needs n x 6 normals");
  const double* rn[1] = {replay_normals};
  return run_resample(&h, 1, uniforms, n_uniforms, rn);
}

int mcl_resample_prepare(mcl_handle* h, int64_t* n_uniforms) {
  if (!h || !n_uniforms) return MCL_ERR_INVALID;
  if (!h->have_lw) return fail(h, MCL_ERR_STATE, "resample_prepare: no weights (call an update first)");
  RET_IF(set_device(h));
  int rc;
  *n_uniforms = uniforms_needed(h, &rc);
  return rc;
}

int mcl_group_resample(mcl_handle** shards, int32_t ns, const double* uniforms, int64_t n_uniforms,
                       const double* const* replay_normals) {
  if (!shards || ns < 1) return MCL_ERR_INVALID;
  for (int s = 0; s < ns; ++s)
    if (!shards[s] || shards[s]->world != ns || shards[s]->rank != s)
      return fail(shards[0], MCL_ERR_INVALID, "group_resample: shards must be ranks 0..n-1 of one world");
  return run_resample(shards, ns, uniforms, n_uniforms, replay_normals);
}

int mcl_mean_cov(mcl_handle* h, double mean6[6], double* yaw_mean, double cov9[9]) {
  if (!h || !mean6 || !cov9) return MCL_ERR_INVALID;
  if (h->world > 1 && !h->comm) return fail(h, MCL_ERR_STATE, "mean_cov: multi-shard handle needs a communicator");
  RET_IF(run_mean_cov_async(&h, 1));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  finish_mean_cov(h, mean6, yaw_mean, cov9);
  return MCL_OK;
}

int mcl_mean_cov_async(mcl_handle* h) {
  if (!h) return MCL_ERR_INVALID;
  if (h->world > 1 && !h->comm) return fail(h, MCL_ERR_STATE, "mean_cov: multi-shard handle needs a communicator");
  return run_mean_cov_async(&h, 1);
}

int mcl_group_mean_cov(mcl_handle** shards, int32_t ns, double mean6[6], double* yaw_mean, double cov9[9]) {
  if (!shards || ns < 1 || !mean6 || !cov9) return MCL_ERR_INVALID;
  RET_IF(run_mean_cov_async(shards, ns));
  for (int s = 0; s < ns; ++s) {
    RET_IF(set_device(shards[s]));
    HIPCHK(shards[s], hipStreamSynchronize(shards[s]->stream));
  }
  finish_mean_cov(shards[0], mean6, yaw_mean, cov9);
  return MCL_OK;
}

int mcl_last_mean_cov(mcl_handle* h, double mean6[6], double* yaw_mean, double cov9[9]) {
  if (!h || !mean6 || !cov9) return MCL_ERR_INVALID;
  if (!h->have_meancov) return fail(h, MCL_ERR_STATE, "last_mean_cov: nothing computed yet");
  RET_IF(set_device(h));
  RET_IF(flush_pending_moments(h));   // (one process per GPU: the last fused step's sums may still be per shard -- a collective)
  HIPCHK(h, hipStreamSynchronize(h->stream));
  finish_mean_cov(h, mean6, yaw_mean, cov9);
  return MCL_OK;
}

int mcl_mean_history(mcl_handle* h, int64_t last_k, double* mean6_out) {
  if (!h || !mean6_out || last_k < 1) return MCL_ERR_INVALID;
  if (last_k > h->mean_count || last_k > MEAN_RING) return fail(h, MCL_ERR_INVALID, "mean_history: not that many results kept");
  RET_IF(set_device(h));
  RET_IF(flush_pending_moments(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (long long k = 0; k < last_k; ++k) {
    double yaw, cov9[9];
    finish_mean_cov(h, mean6_out + 6 * k, &yaw, cov9, h->mean_count - last_k + k);
  }
  return MCL_OK;
}

int mcl_get_poses(mcl_handle* h, double* pose7) {
  if (!h || !pose7) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  if (!h->pose7) HIPCHK(h, hipMalloc(&h->pose7, sizeof(double) * 7 * (size_t)h->n));
  k_poses<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(state_ptrs(h->state[h->cur], h->n), h->n, h->pose7);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(pose7, h->pose7, sizeof(double) * 7 * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MCL_OK;
}

int mcl_get_particles(mcl_handle* h, double* soa, double* w) {
  if (!h || !soa) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  HIPCHK(h, hipMemcpyAsync(soa, h->state[h->cur], sizeof(double) * 6 * (size_t)h->n, hipMemcpyDeviceToHost,
                           h->stream));
  if (w) {
    if (!h->have_cdf) return fail(h, MCL_ERR_STATE, "get_particles: weights exist only after a resample");
    if (!h->wnorm) HIPCHK(h, hipMalloc(&h->wnorm, sizeof(double) * (size_t)h->n));
    k_normalised_weights<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->q, h->n, h->totals, h->world, h->qshift_cur, h->wnorm);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(w, h->wnorm, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MCL_OK;
}

int mcl_set_particles(mcl_handle* h, const double* soa) {
  if (!h || !soa) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  RET_IF(cancel_state_gather(h));
  h->uni_valid = false;
  h->visit_ready = false;
  HIPCHK(h, hipMemcpyAsync(h->state[h->cur], soa, sizeof(double) * 6 * (size_t)h->n, hipMemcpyHostToDevice,
                           h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MCL_OK;
}

int mcl_get_log_weights(mcl_handle* h, double* lw) {
  if (!h || !lw) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  HIPCHK(h, hipMemcpyAsync(lw, h->lw, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MCL_OK;
}

int mcl_set_log_weights(mcl_handle* h, const double* lw, int32_t weight_mode) {
  if (!h || !lw || weight_mode < 0 || weight_mode > 2) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  HIPCHK(h, hipMemcpyAsync(h->lw, lw, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->weight_mode = weight_mode;
  h->have_lw = true;
  h->max_valid = false;
  h->residual_k = -1;
  return MCL_OK;
}

int mcl_get_last_indices(mcl_handle* h, int32_t* idx) {
  if (!h || !idx) return MCL_ERR_INVALID;
  if (!h->have_cdf && !h->idx_explicit) return fail(h, MCL_ERR_STATE, "get_last_indices: no resample yet");
  RET_IF(set_device(h));
  if (!h->idx) HIPCHK(h, hipMalloc(&h->idx, sizeof(int) * (size_t)h->n));
  if (!h->idx_explicit) RET_IF(ensure_global_cdf(h));
  if (!h->idx_explicit) {
    k_indices<<<grid_for(h->n), MCL_BLOCK, 0, h->stream>>>(h->ncum, h->ng, h->goff, h->n, h->idx);
    HIPCHK(h, hipGetLastError());
  }
  HIPCHK(h, hipMemcpyAsync(idx, h->idx, sizeof(int) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MCL_OK;
}

int mcl_get_last_offspring_cdf(mcl_handle* h, uint32_t* ncum) {
  if (!h || !ncum) return MCL_ERR_INVALID;
  if (!h->have_cdf) return fail(h, MCL_ERR_STATE, "get_last_offspring_cdf: no resample yet");
  RET_IF(set_device(h));
  RET_IF(ensure_global_cdf(h));
  HIPCHK(h, hipMemcpyAsync(ncum, h->ncum, sizeof(u32) * (size_t)h->ng, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MCL_OK;
}

int mcl_get_fixed_weights(mcl_handle* h, uint64_t* q, uint64_t* total) {
  if (!h || !q) return MCL_ERR_INVALID;
  if (!h->have_cdf) return fail(h, MCL_ERR_STATE, "get_fixed_weights: no resample yet");
  RET_IF(set_device(h));
  HIPCHK(h, hipMemcpyAsync(q, h->q, sizeof(u64) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  std::vector<u64> t(h->world);
  HIPCHK(h, hipMemcpyAsync(t.data(), h->totals, sizeof(u64) * (size_t)h->world, hipMemcpyDeviceToHost, h->stream));
  u64 shift = 0;   // (a shard of several processes keeps its weights at its own exponent: mcl_resample.h, k_shift_scan)
  if (h->qshift_cur) HIPCHK(h, hipMemcpyAsync(&shift, h->qshift_cur, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (shift)
    for (long long i = 0; i < h->n; ++i) q[i] = shift >= 64 ? 0 : q[i] >> shift;
  if (total) {
    u64 T = 0;
    for (u64 v : t) T += v;
    *total = T;
  }
  return MCL_OK;
}

namespace {
// the fused step of one handle (mcl_step_mbes; with a landmark observation: mcl_step_mbes_landmarks)
int step_mbes_impl(mcl_handle* h, const mcl_odom* odom, double dt, const float* ranges, const float* beam_angles, int32_t B,
                   double sigma, double r_max, const double sensor_offset[6], const LandmarkObs* lm, const char* who) {
  if (!h || !odom || !ranges || !beam_angles) return MCL_ERR_INVALID;
  const std::string w(who);
  if (h->cfg.rng_mode != MCL_RNG_NATIVE) return fail(h, MCL_ERR_INVALID, w + ": NATIVE rng only");
  if (h->world > 1 && !h->comm) return fail(h, MCL_ERR_STATE, w + ": multi-shard handle needs mcl_comm_init");
  if (B < 1 || !(sigma > 0.0) || !(r_max > 0.0)) return fail(h, MCL_ERR_INVALID, w + ": bad argument");
  if (h->map_kind < 0) return fail(h, MCL_ERR_STATE, w + ": no map (call mcl_set_map_grid/mesh first)");
  RET_IF(set_device(h));
  // predict writes the MBES pose records of the new state in the same pass (the map and sensor offset are known here)
  // (the beam table first: the group classification in that kernel follows the two extreme beams)
  RET_IF(upload_beams(h, ranges, beam_angles, B));
  if (lm) RET_IF(landmarks_prepare(h, *lm, who, true));   // (after upload_beams: it forgets detections an earlier call left waiting)
  MbesArgs pa;
  RET_IF(launch_mbes(h, true, B, sigma, r_max, sensor_offset, h->lw, nullptr, 0, 0, false, &pa));
  bool pose_done = false;
  const bool sys = h->cfg.resample_scheme == MCL_RESAMPLE_SYSTEMATIC || h->cfg.resample_scheme == MCL_RESAMPLE_NAIVE;
  // (systematic scheme: the gather of this call substitutes z, roll, pitch -- the predict kernel does not store them)
  RET_IF(do_predict(h, odom, dt, nullptr, &pa, &pose_done, sys));
  int rc_u = h->fault_step ? fail(h, MCL_ERR_STATE, w + ": injected fault after predict") : start_state_gather(h);
  if (rc_u == MCL_OK) rc_u = launch_mbes(h, true, B, sigma, r_max, sensor_offset, h->lw, nullptr, 0, 0, pose_done);
  if (rc_u == MCL_OK) {
    h->weight_mode = MCL_WEIGHT_LOG_SHIFT;
    h->have_lw = true;
    h->residual_k = -1;
    // the landmark likelihood of the same ping on top (BASELINE config 5): reads the state the predict left (z, roll,
    // pitch from the odometry when that kernel did not store them), leaves max lw in the second slot set
    if (lm) rc_u = landmarks_launch(h, *lm, true, true);
  }
  if (rc_u != MCL_OK) {
    const std::string keep = h->err;
    (void)cancel_state_gather(h);
    (void)materialise_uniform(h);
    h->err = keep;
    return rc_u;
  }
  // resample; the gather pass also accumulates the sums of update_loc_pose of the new state
  rc_u = run_resample(&h, 1, nullptr, 0, nullptr, sys);
  if (rc_u != MCL_OK) {
    const std::string keep = h->err;
    (void)materialise_uniform(h);
    h->err = keep;
    return rc_u;
  }
  if (sys)
    RET_IF(collect_fused_moments(&h, 1));
  else
    RET_IF(run_mean_cov_async(&h, 1));
  return MCL_OK;
}

int group_step_mbes_impl(mcl_handle** shards, int32_t ns, const mcl_odom* odom, double dt, const float* ranges,
                         const float* beam_angles, int32_t B, double sigma, double r_max, const double sensor_offset[6],
                         const LandmarkObs* lm, const char* who) {
  if (!shards || ns < 1 || !odom || !ranges || !beam_angles) return MCL_ERR_INVALID;
  const std::string w(who);
  for (int s = 0; s < ns; ++s)
    if (!shards[s] || shards[s]->world != ns || shards[s]->rank != s)
      return fail(shards[0], MCL_ERR_INVALID, w + ": shards must be ranks 0..n-1 of one world");
  if (B < 1 || !(sigma > 0.0) || !(r_max > 0.0)) return fail(shards[0], MCL_ERR_INVALID, w + ": bad argument");
  for (int s = 0; s < ns; ++s) {
    mcl_handle* h = shards[s];
    if (h->cfg.rng_mode != MCL_RNG_NATIVE) return fail(h, MCL_ERR_INVALID, w + ": NATIVE rng only");
    if (h->map_kind < 0) return fail(h, MCL_ERR_STATE, w + ": no map (call mcl_set_map_grid/mesh first)");
    if (h->cfg.resample_scheme != MCL_RESAMPLE_SYSTEMATIC && h->cfg.resample_scheme != MCL_RESAMPLE_NAIVE)
      return fail(h, MCL_ERR_UNSUPPORTED, w + ": only the systematic scheme is sharded");
    if (lm && !h->landmarks) return fail(h, MCL_ERR_STATE, w + ": no feature map (call mcl_set_landmarks first)");
  }
  // everything that can fail before a kernel is queued, for EVERY shard first: a later shard's failure must not find
  // earlier shards with a predict in flight whose z / roll / pitch stores were deferred to the gather
  for (int s = 0; s < ns; ++s) {
    mcl_handle* h = shards[s];
    RET_IF(set_device(h));
    RET_IF(upload_beams(h, ranges, beam_angles, B));
    if (lm) RET_IF(landmarks_prepare(h, *lm, who, true));
  }
  for (int s = 0; s < ns; ++s) {
    mcl_handle* h = shards[s];
    // the same fused front half as mcl_step_mbes: predict writes the pose records, the sweep leaves max lw in the slots
    int rc = set_device(h);
    MbesArgs pa;
    if (rc == MCL_OK) rc = launch_mbes(h, true, B, sigma, r_max, sensor_offset, h->lw, nullptr, 0, 0, false, &pa);
    bool pose_done = false;
    if (rc == MCL_OK) rc = do_predict(h, odom, dt, nullptr, &pa, &pose_done, true);
    if (rc == MCL_OK && h->fault_step) rc = fail(h, MCL_ERR_STATE, w + ": injected fault after predict");
    if (rc == MCL_OK) rc = launch_mbes(h, true, B, sigma, r_max, sensor_offset, h->lw, nullptr, 0, 0, pose_done);
    if (rc == MCL_OK) {
      h->weight_mode = MCL_WEIGHT_LOG_SHIFT;
      h->have_lw = true;
      h->residual_k = -1;
      if (lm) rc = landmarks_launch(h, *lm, true, true);
    }
    if (rc != MCL_OK) {
      const std::string keep = h->err;
      for (int t = 0; t <= s; ++t) (void)materialise_uniform(shards[t]);
      h->err = keep;
      return rc;
    }
  }
  const int rc = run_resample(shards, ns, nullptr, 0, nullptr, true);
  if (rc != MCL_OK) {
    const std::string keep = shards[0]->err;
    for (int t = 0; t < ns; ++t) (void)materialise_uniform(shards[t]);
    shards[0]->err = keep;
    return rc;
  }
  return collect_fused_moments(shards, ns);
}
}  // namespace

int mcl_step_mbes(mcl_handle* h, const mcl_odom* odom, double dt, const float* ranges, const float* beam_angles,
                  int32_t B, double sigma, double r_max, const double sensor_offset[6]) {
  return step_mbes_impl(h, odom, dt, ranges, beam_angles, B, sigma, r_max, sensor_offset, nullptr, "step_mbes");
}

int mcl_step_mbes_landmarks(mcl_handle* h, const mcl_odom* odom, double dt, const float* ranges, const float* beam_angles,
                            int32_t B, double sigma, double r_max, const double sensor_offset[6], const double* det_xyz,
                            int32_t n_det, double lm_sigma, int32_t k, double gate, const double lm_sensor_offset[6]) {
  const LandmarkObs o = {det_xyz, n_det, lm_sigma, k, gate, lm_sensor_offset};
  return step_mbes_impl(h, odom, dt, ranges, beam_angles, B, sigma, r_max, sensor_offset, &o, "step_mbes_landmarks");
}

int mcl_group_step_mbes(mcl_handle** shards, int32_t ns, const mcl_odom* odom, double dt, const float* ranges,
                        const float* beam_angles, int32_t B, double sigma, double r_max, const double sensor_offset[6]) {
  return group_step_mbes_impl(shards, ns, odom, dt, ranges, beam_angles, B, sigma, r_max, sensor_offset, nullptr,
                              "group_step_mbes");
}

int mcl_group_step_mbes_landmarks(mcl_handle** shards, int32_t ns, const mcl_odom* odom, double dt, const float* ranges,
                                  const float* beam_angles, int32_t B, double sigma, double r_max,
                                  const double sensor_offset[6], const double* det_xyz, int32_t n_det, double lm_sigma,
                                  int32_t k, double gate, const double lm_sensor_offset[6]) {
  const LandmarkObs o = {det_xyz, n_det, lm_sigma, k, gate, lm_sensor_offset};
  return group_step_mbes_impl(shards, ns, odom, dt, ranges, beam_angles, B, sigma, r_max, sensor_offset, &o,
                              "group_step_mbes_landmarks");
}

int mcl_exchange_plan(int32_t world, const uint32_t* lost, const uint32_t* surplus, int32_t rank, uint32_t* send_off,
                      uint32_t* send_cnt, uint32_t* recv_off, uint32_t* recv_cnt) {
  return exchange_plan_impl(world, lost, surplus, rank, send_off, send_cnt, recv_off, recv_cnt);
}

int mcl_exchange_stats(mcl_handle* h, int64_t* states_sent, int64_t* lost_slots, int32_t reset) {
  if (!h) return MCL_ERR_INVALID;
  if (states_sent) *states_sent = (int64_t)h->ex_sent;
  if (lost_slots) *lost_slots = (int64_t)h->ex_lost;
  if (reset) h->ex_sent = h->ex_lost = 0;
  return MCL_OK;
}

int mcl_exchange_ops(mcl_handle* h, int64_t* p2p_ops, int64_t* resamples, int32_t reset) {
  if (!h) return MCL_ERR_INVALID;
  if (p2p_ops) *p2p_ops = (int64_t)h->ex_ops;
  if (resamples) *resamples = (int64_t)h->ex_rounds;
  if (reset) h->ex_ops = h->ex_rounds = 0;
  return MCL_OK;
}

int mcl_sync(mcl_handle* h) {
  if (!h) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MCL_OK;
}

int mcl_resample_indices(int32_t scheme, const double* weights, int64_t n, const double* uniforms,
                         int64_t n_uniforms, int32_t device, int32_t* out) {
  if (!weights || !out || n < 1) return MCL_ERR_INVALID;
  mcl_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.n_particles = n;
  cfg.device = device;
  cfg.resample_scheme = scheme;
  cfg.rng_mode = MCL_RNG_REPLAY;
  mcl_handle* h = nullptr;
  int rc = mcl_create(&cfg, &h);
  if (rc != MCL_OK) return rc;
  rc = mcl_set_log_weights(h, weights, MCL_WEIGHT_LINEAR);
  if (rc == MCL_OK) {
    if (scheme != MCL_RESAMPLE_SYSTEMATIC && scheme != MCL_RESAMPLE_NAIVE) {
      int64_t need = 0;
      rc = mcl_resample_prepare(h, &need);
      if (rc == MCL_OK) rc = alt_indices(h, uniforms, n_uniforms);
      if (rc == MCL_OK) rc = mcl_get_last_indices(h, out);
      if (rc != MCL_OK) g_create_err = h->err;
    } else if (!uniforms || n_uniforms < 1 || !(uniforms[0] >= 0.0 && uniforms[0] < 1.0)) {
      g_create_err = "resample_indices: systematic needs one uniform in [0,1)";
      rc = MCL_ERR_INVALID;
    } else {
      uint64_t u53 = (uint64_t)std::floor(uniforms[0] * 9007199254740992.0);
      if (scheme == MCL_RESAMPLE_NAIVE) u53 |= MCL_U53_NAIVE;
      rc = phase_quantise(h, true);
      if (rc == MCL_OK) rc = phase_cdf(h, u53);
      if (rc == MCL_OK) {
        h->have_cdf = true;
        h->cdf_global = true;
        rc = mcl_get_last_indices(h, out);
      }
      if (rc != MCL_OK) g_create_err = h->err;
    }
  } else {
    g_create_err = h->err;
  }
  mcl_destroy(h);
  return rc;
}

int mcl_comm_unique_id(char id[128]) {
  if (!id) return MCL_ERR_INVALID;
  ncclUniqueId uid;
  static_assert(sizeof(ncclUniqueId) <= 128, "unique id size");
  if (ncclGetUniqueId(&uid) != ncclSuccess) {
    g_create_err = "ncclGetUniqueId failed";
    return MCL_ERR_COMM;
  }
  memset(id, 0, 128);
  memcpy(id, &uid, sizeof uid);
  return MCL_OK;
}

namespace {
// wait for an event with a deadline; 0 = done, 1 = timed out, negative = HIP error
int wait_event_ms(hipEvent_t ev, int timeout_ms) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return 0;
    if (e != hipErrorNotReady) return -1;
    if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms)
      return 1;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}
void comm_teardown(mcl_handle* h, bool abort) {
  if (h->comm2) (void)(abort ? ncclCommAbort(h->comm2) : ncclCommDestroy(h->comm2));
  if (h->comm) (void)(abort ? ncclCommAbort(h->comm) : ncclCommDestroy(h->comm));
  h->comm2 = nullptr;
  h->comm = nullptr;
  h->gather_inflight = false;
}
}  // namespace

int mcl_comm_init_ex(mcl_handle* h, const char id[128], uint32_t flags) {
  if (!h || !id) return MCL_ERR_INVALID;
  if (h->world < 2 && !h->env_force_comm) return MCL_OK;  // MCL_FORCE_COMM=1: test hook, 1-rank communicator
  if (h->comm) return fail(h, MCL_ERR_STATE, "comm_init: communicator exists (mcl_comm_shutdown first)");
  RET_IF(set_device(h));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  NCCLCHK(h, ncclCommInitRank(&h->comm, h->world, uid, h->rank));
  if (h->exch_allgather && !h->state_glob) HIPCHK(h, hipMalloc(&h->state_glob, sizeof(double) * 6 * (size_t)h->ng));
  // second communicator + stream for the overlapped state all-gather (MCL_EXCHANGE=allgather only: the O(n)
  // exchange ships a few per cent of a shard and has nothing worth hiding); optional
  const bool overlap = h->exch_allgather && !(flags & MCL_COMM_NO_OVERLAP) && !h->env_no_overlap;
  if (overlap && ncclCommSplit(h->comm, 0, h->rank, &h->comm2, nullptr) == ncclSuccess && h->comm2) {
    if (!h->comm_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    if (!h->ev_state_ready) HIPCHK(h, hipEventCreateWithFlags(&h->ev_state_ready, hipEventDisableTiming));
    if (!h->ev_gather_done) HIPCHK(h, hipEventCreateWithFlags(&h->ev_gather_done, hipEventDisableTiming));
  } else {
    h->comm2 = nullptr;
  }
  return MCL_OK;
}

int mcl_comm_init(mcl_handle* h, const char id[128]) { return mcl_comm_init_ex(h, id, 0u); }

int mcl_comm_ranks(mcl_handle* h, int32_t* ranks, int32_t* overlap) {
  if (!h || !ranks) return MCL_ERR_INVALID;
  if (overlap) *overlap = h->comm2 ? 1 : 0;
  if (!h->comm) {
    *ranks = 1;
    return MCL_OK;
  }
  RET_IF(set_device(h));
  // every rank contributes 1: the sum is the number of ranks RCCL really connected
  int* d = (int*)(h->totals + h->world);  // scratch word behind the shard totals
  const int one = 1;
  HIPCHK(h, hipMemcpyAsync(d, &one, sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  NCCLCHK(h, ncclAllReduce(d, d, 1, ncclInt32, ncclSum, h->comm, h->stream));
  int got = 0;
  HIPCHK(h, hipMemcpyAsync(&got, d, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  *ranks = got;
  return MCL_OK;
}

int mcl_comm_selftest(mcl_handle* h, int32_t timeout_ms) {
  if (!h) return MCL_ERR_INVALID;
  if (!h->comm) return MCL_OK;
  if (timeout_ms < 1) timeout_ms = 1;
  RET_IF(set_device(h));
  // the exact concurrency pattern of mcl_step_mbes: the 6-array state all-gather on the second
  // communicator/stream while the first communicator runs its all-reduce + all-gathers, three rounds
  hipEvent_t done = nullptr;
  HIPCHK(h, hipEventCreateWithFlags(&done, hipEventDisableTiming));
  int rc = MCL_OK;
  for (int round = 0; round < 3 && rc == MCL_OK; ++round) {
    rc = start_state_gather(h);
    if (rc != MCL_OK) break;
    ncclResult_t e = ncclAllReduce(h->scal + 24, h->scal + 24, 1, ncclDouble, ncclMax, h->comm, h->stream);
    if (e == ncclSuccess) e = ncclAllGather(h->totals + h->rank, h->totals, 1, ncclUint64, h->comm, h->stream);
    if (e == ncclSuccess && h->exch_allgather)
      e = ncclAllGather(h->ncum + h->goff, h->ncum, (size_t)h->n, ncclUint32, h->comm, h->stream);
    if (e == ncclSuccess && !h->exch_allgather && h->world > 1) {
      // the O(n) exchange's pattern: the hand-over records all-gathered, then grouped point-to-point transfers
      // (here: one word to the next rank, one from the previous)
      RET_IF(alloc_lsx(h));
      e = ncclAllGather(h->lsx + 4 * (size_t)h->rank, h->lsx, 4, ncclUint64, h->comm, h->stream);
      if (e == ncclSuccess) e = ncclGroupStart();
      if (e == ncclSuccess) e = ncclSend(h->totals + h->rank, 1, ncclUint64, (h->rank + 1) % h->world, h->comm, h->stream);
      if (e == ncclSuccess) e = ncclRecv(h->totals + h->world, 1, ncclUint64, (h->rank + h->world - 1) % h->world, h->comm, h->stream);
      if (e == ncclSuccess) e = ncclGroupEnd();
    }
    if (e != ncclSuccess) {
      h->err = std::string("comm_selftest: ") + ncclGetErrorString(e);
      rc = MCL_ERR_COMM;
      break;
    }
    if (h->gather_inflight) {
      (void)hipStreamWaitEvent(h->stream, h->ev_gather_done, 0);
      h->gather_inflight = false;
    }
    (void)hipEventRecord(done, h->stream);
    const int w = wait_event_ms(done, timeout_ms);
    if (w != 0) {
      h->err = w > 0 ? "comm_selftest: collectives did not complete before the deadline (communicators aborted)"
                     : "comm_selftest: HIP error while waiting";
      comm_teardown(h, true);
      rc = MCL_ERR_COMM;
    }
  }
  (void)hipEventDestroy(done);
  return rc;
}

int mcl_comm_shutdown(mcl_handle* h, int32_t abort) {
  if (!h) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  if (!abort) RET_IF(flush_pending_moments(h));   // (while the communicator is still there)
  if (!abort && h->stream) HIPCHK(h, hipStreamSynchronize(h->stream));
  if (!abort && h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  comm_teardown(h, abort != 0);
  (void)flush_pending_moments(h);   // (aborted with an entry open: it is marked NaN)
  return MCL_OK;
}

int mcl_mbes_last_path(mcl_handle* h, int32_t* path, int64_t* handed_over, int64_t* deferred_groups) {
  if (!h) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  int cnt[2] = {0, 0};  // groups the fast kernel left to the general one; particles the sweep handed over
  HIPCHK(h, hipMemcpy(cnt, h->ctrl + CTRL_WORK, sizeof cnt, hipMemcpyDeviceToHost));
  if (path) *path = h->sweep_now ? 1 : (h->slice_now ? 2 : 0);
  if (handed_over) *handed_over = (h->sweep_now || h->slice_now) ? cnt[1] : 0;
  if (deferred_groups) {
    *deferred_groups = cnt[0];
    if (h->slice_now && h->slice_group_ran && h->slice_loose) {
      // the fan slice over groups of spatial neighbours: groups it left to the per-particle kernel
      int loose = 0;
      HIPCHK(h, hipMemcpy(&loose, h->ctrl + CTRL_LOOSE, sizeof loose, hipMemcpyDeviceToHost));
      *deferred_groups = loose;
    } else if (h->slice_now) {
      *deferred_groups = -1;   // (no groups: every particle cast on its own)
    }
  }
  return MCL_OK;
}

int mcl_mbes_last_handover(mcl_handle* h, int64_t* by_slice, int64_t* by_traversal) {
  if (!h) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  int first = 0, second = 0;
  HIPCHK(h, hipMemcpy(&first, h->ctrl + CTRL_DEFER, sizeof first, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(&second, h->ctrl + CTRL_DEFER2, sizeof second, hipMemcpyDeviceToHost));
  const bool staged = h->sweep_now && h->handover_slice_now;
  if (!h->sweep_now && !h->slice_now) first = 0;
  if (by_slice) *by_slice = staged ? first - second : 0;
  if (by_traversal) *by_traversal = staged ? second : first;
  return MCL_OK;
}

int mcl_mbes_visit_order(mcl_handle* h, uint32_t* slots, int32_t* sorted) {
  if (!h || !slots) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (sorted) *sorted = h->pose_visit ? 1 : 0;
  if (!h->pose_visit || !h->pose_dev) {
    for (long long i = 0; i < h->n; ++i) slots[i] = (uint32_t)i;
    return MCL_OK;
  }
  std::vector<MbesPose> rec((size_t)h->n);
  HIPCHK(h, hipMemcpy(rec.data(), h->pose_dev, sizeof(MbesPose) * (size_t)h->n, hipMemcpyDeviceToHost));
  for (long long i = 0; i < h->n; ++i) slots[i] = rec[(size_t)i].slot;
  return MCL_OK;
}

int mcl_timing_enable(mcl_handle* h, int32_t on) {
  if (!h) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  t_collect(h);
  h->timing = on != 0;
  return MCL_OK;
}

int mcl_timing_get(mcl_handle* h, mcl_timing* out) {
  if (!h || !out) return MCL_ERR_INVALID;
  RET_IF(set_device(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  t_collect(h);
  *out = h->tacc;
  memset(&h->tacc, 0, sizeof h->tacc);
  return MCL_OK;
}

}  // extern "C"

#ifdef SWEEP_TIMELINE
// debug builds only (tools/sweep_timeline.py): the per-wave clock records of the last sweep launch
extern "C" int mcl_debug_sweep_timeline(unsigned long long* out, int n_waves) {
  if (n_waves > SWEEP_TL_WAVES) n_waves = SWEEP_TL_WAVES;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sweep_tl), (size_t)n_waves * 6 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

