// sba_api.hip -- C ABI (include/sba_hip.h) over the per-camera-model engines.  gfx950 only; no CPU fallback:
// every entry point fails with SBA_ERR_NO_DEVICE / SBA_ERR_HIP when the GPU is not usable.
#include "sba_common.hpp"

using namespace sba_host;

// factories of the two engine translation units (sba_engine_ncp11.hip: the reference's 11-parameter radial camera,
// sba_engine_ncp13.hip: 13 parameters, radial + tangential)
sba_host::EngineBase* sba_make_engine_ncp11(int dtype);
sba_host::EngineBase* sba_make_engine_ncp13(int dtype);
int sba_rows_call_ncp11(int dtype, bool project, int device, int64_t n, const double* pts, const double* other, double* out);
int sba_rows_call_ncp13(int dtype, bool project, int device, int64_t n, const double* pts, const double* other, double* out);

// ============================================================================================== C ABI
struct sba_handle {
  int dtype;
  std::unique_ptr<EngineBase> eng;
  std::string err;
};

namespace {

template <typename F>
int guarded(sba_handle* h, F&& f) {
  ArenaScope arena_scope(h && h->eng ? &h->eng->arena : nullptr);    // buffers allocated during the call belong to the handle
  try {
    int rc = f();
    if (rc && h) h->err = h->eng ? h->eng->err : h->err;
    return rc;
  } catch (const HipError& e) {
    char buf[512];
    const char* base = strrchr(e.file, '/');
    snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d: %s", (int)e.e, hipGetErrorString(e.e), base ? base + 1 : e.file, e.line, e.what);
    if (h) h->err = buf; else g_last_error = buf;
    return SBA_ERR_HIP;
  } catch (const RcclError& e) {
    char buf[512];
    Rccl& r = Rccl::get();
    snprintf(buf, sizeof buf, "RCCL error %d (%s) at sba_engine.hpp:%d: %s", e.code, r.error_string ? r.error_string(e.code) : "?", e.line, e.what);
    if (h) h->err = buf; else g_last_error = buf;
    return SBA_ERR_HIP;
  } catch (const std::exception& e) {
    if (h) h->err = e.what(); else g_last_error = e.what();
    return SBA_ERR_INVALID;
  }
}

int check_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) { g_last_error = "no HIP device is visible (libsba_hip needs an MI355X / gfx950 GPU)"; return SBA_ERR_NO_DEVICE; }
  if (device < 0 || device >= n) { g_last_error = "device ordinal out of range"; return SBA_ERR_NO_DEVICE; }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) { g_last_error = "hipGetDeviceProperties failed"; return SBA_ERR_NO_DEVICE; }
  if (std::string(p.gcnArchName).rfind("gfx950", 0) != 0) {
    g_last_error = std::string("device is ") + p.gcnArchName + ", libsba_hip is built for gfx950 only";
    return SBA_ERR_NO_DEVICE;
  }
  return SBA_OK;
}

}  // namespace

extern "C" {

int sba_abi_version(void) { return SBA_ABI_VERSION; }

int sba_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* sba_last_error(const sba_handle* h) { return h ? h->err.c_str() : g_last_error.c_str(); }

static int n_cam_params_of(int model) { return model == SBA_CAM_RADIAL_TANGENTIAL ? 13 : 11; }
static int rows_dispatch(int ncp, int dtype, bool project, int device, int64_t n, const double* pts, const double* other, double* out) {
  return ncp == 13 ? sba_rows_call_ncp13(dtype, project, device, n, pts, other, out)
                   : sba_rows_call_ncp11(dtype, project, device, n, pts, other, out);
}

int sba_rotate(int device, int dtype, int64_t n, const double* points, const double* rot_vecs, double* out) {
  if (n < 0 || (n > 0 && (!points || !rot_vecs || !out))) { g_last_error = "null argument"; return SBA_ERR_INVALID; }
  int rc = check_device(device);
  if (rc) return rc;
  return guarded(nullptr, [&] { return rows_dispatch(11, dtype, false, device, n, points, rot_vecs, out); });
}

int sba_project(int device, int dtype, int64_t n, const double* points, const double* cam_rows, double* uv_out) {
  return sba_project_model(device, dtype, SBA_CAM_RADIAL, n, points, cam_rows, uv_out);
}

int sba_project_model(int device, int dtype, int cam_model, int64_t n, const double* points, const double* cam_rows, double* uv_out) {
  if (n < 0 || (n > 0 && (!points || !cam_rows || !uv_out))) { g_last_error = "null argument"; return SBA_ERR_INVALID; }
  if (cam_model != SBA_CAM_RADIAL && cam_model != SBA_CAM_RADIAL_TANGENTIAL) { g_last_error = "unknown camera model"; return SBA_ERR_INVALID; }
  int rc = check_device(device);
  if (rc) return rc;
  return guarded(nullptr, [&] { return rows_dispatch(n_cam_params_of(cam_model), dtype, true, device, n, points, cam_rows, uv_out); });
}

int sba_create(const sba_problem_desc* desc, sba_handle** out) {
  if (!desc || !out) { g_last_error = "null argument"; return SBA_ERR_INVALID; }
  *out = nullptr;
  if (desc->dtype != SBA_F64 && desc->dtype != SBA_F32) { g_last_error = "dtype must be SBA_F64 or SBA_F32"; return SBA_ERR_INVALID; }
  if (desc->cam_model != SBA_CAM_RADIAL && desc->cam_model != SBA_CAM_RADIAL_TANGENTIAL) { g_last_error = "unknown camera model"; return SBA_ERR_INVALID; }
  int rc = check_device(desc->device);
  if (rc) return rc;
  auto h = std::make_unique<sba_handle>();
  h->dtype = desc->dtype;
  rc = guarded(nullptr, [&] {
    std::unique_ptr<EngineBase> e(desc->cam_model == SBA_CAM_RADIAL_TANGENTIAL ? sba_make_engine_ncp13(desc->dtype)
                                                                                : sba_make_engine_ncp11(desc->dtype));
    ArenaScope sc(&e->arena);
    e->init(*desc);
    h->eng = std::move(e);
    return (int)SBA_OK;
  });
  if (rc) return rc;
  *out = h.release();
  return SBA_OK;
}

int sba_destroy(sba_handle* h) {
  if (!h) return SBA_OK;
  delete h;
  return SBA_OK;
}

int sba_upload(sba_handle* h, const double* cams, const double* points, const double* uv, const int64_t* cam_idx,
               const int64_t* pt_idx, const double* weights) {
  if (!h) return SBA_ERR_INVALID;
  if (!cams || !points || ((!uv || !cam_idx || !pt_idx))) { h->err = "null argument"; return SBA_ERR_INVALID; }
  return guarded(h, [&] { return h->eng->upload(cams, points, uv, cam_idx, pt_idx, weights); });
}

int sba_set_params(sba_handle* h, const double* x) {
  if (!h || !x) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->set_params_x(x); });
}

int sba_get_params(sba_handle* h, double* cams_out, double* points_out) {
  if (!h) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->get_params(cams_out, points_out); });
}

int sba_get_transform(sba_handle* h, double* theta12_out) {
  if (!h || !theta12_out) return SBA_ERR_INVALID;
  return h->eng->get_transform(theta12_out);
}

int sba_lm_get_step(sba_handle* h, double* delta_c_out) {
  if (!h || !delta_c_out) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->get_step(delta_c_out); });
}

int sba_get_gradient(sba_handle* h, double* gc_out, double* gp_out) {
  if (!h) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->get_gradient(gc_out, gp_out); });
}

int sba_residual(sba_handle* h, const double* x, double* r_out, double* cost_out) {
  if (!h) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->residual(x, r_out, cost_out); });
}

int sba_residual_jacobian(sba_handle* h, const double* x, double* r_out, double* Jc_out, double* Jp_out) {
  if (!h) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->residual_jacobian(x, r_out, Jc_out, Jp_out); });
}

int sba_solve_lm(sba_handle* h, const sba_lm_opts* opts, double* cams_out, double* points_out, sba_lm_report* report,
                 sba_lm_iter_log* log, int32_t log_capacity, int32_t* log_rows) {
  if (!h || !opts) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->solve(opts, cams_out, points_out, report, log, log_capacity, log_rows); });
}

int64_t sba_lm_exchange_size(const sba_handle* h) { return h ? h->eng->exchange_size() : 0; }

int sba_lm_begin(sba_handle* h, const sba_lm_opts* opts) {
  if (!h || !opts) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->lm_begin(opts); });
}
int sba_lm_linearize(sba_handle* h) {
  if (!h) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->lm_linearize(); });
}
int sba_lm_form_reduced(sba_handle* h, double* exchange_dev) {
  if (!h || !exchange_dev) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->lm_form_reduced(exchange_dev); });
}
int sba_lm_solve_trial(sba_handle* h, const double* exchange_dev, double* scalars_dev) {
  if (!h || !exchange_dev || !scalars_dev) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->lm_solve_trial(const_cast<double*>(exchange_dev), scalars_dev); });
}
int sba_lm_decide(sba_handle* h, const double* scalars_all_dev, int32_t n_ranks, int32_t* status_out,
                  int32_t* accepted_out, sba_lm_iter_log* row_out) {
  if (!h || !scalars_all_dev || n_ranks < 1) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->lm_decide(scalars_all_dev, n_ranks, status_out, accepted_out, row_out); });
}
int sba_lm_decide_async(sba_handle* h, const double* scalars_all_dev, int32_t n_ranks) {
  if (!h || n_ranks < 1 || (n_ranks > 1 && !scalars_all_dev)) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->lm_decide_async(scalars_all_dev, n_ranks); });
}
int sba_lm_poll(sba_handle* h, int32_t* status_out, int32_t* iterations_out) {
  if (!h) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->lm_poll(status_out, iterations_out); });
}

int sba_lm_run(sba_handle* h, int32_t* status_out, int32_t* iterations_out) {
  if (!h) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->lm_run(status_out, iterations_out); });
}
int sba_lm_get_log(sba_handle* h, sba_lm_iter_log* log, int32_t log_capacity, int32_t* log_rows) {
  if (!h || !log_rows) return SBA_ERR_INVALID;
  return h->eng->get_log(log, log_capacity, log_rows);
}
int sba_lm_finish(sba_handle* h, double* cams_out, double* points_out, sba_lm_report* report) {
  if (!h) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->lm_finish(cams_out, points_out, report); });
}

int sba_get_kernel_profile(sba_handle* h, double* total_us_out, int64_t* count_out) {
  if (!h || !total_us_out || !count_out) return SBA_ERR_INVALID;
  return h->eng->get_kernel_profile(total_us_out, count_out);
}

int sba_comm_get_unique_id(uint8_t* id_out) {
  if (!id_out) { g_last_error = "null argument"; return SBA_ERR_INVALID; }
  Rccl& r = Rccl::get();
  if (!r.ok) { g_last_error = r.why; return SBA_ERR_UNSUPPORTED; }
  Rccl::UniqueId uid;
  const int rc = r.get_unique_id(&uid);
  if (rc != 0) { g_last_error = std::string("ncclGetUniqueId failed: ") + r.error_string(rc); return SBA_ERR_HIP; }
  memcpy(id_out, uid.internal, Rccl::ID_BYTES);
  return SBA_OK;
}

int sba_comm_init(sba_handle* h, const uint8_t* id, int32_t rank, int32_t n_ranks) {
  if (!h || !id) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->comm_init(id, rank, n_ranks); });
}

int sba_ipc_export(sba_handle* h, int32_t n_ranks, uint8_t* handle_out) {
  if (!h || !handle_out) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->ipc_export(n_ranks, handle_out); });
}

int sba_ipc_attach(sba_handle* h, int32_t rank, int32_t n_ranks, const uint8_t* handles_all) {
  if (!h || !handles_all) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->ipc_attach(rank, n_ranks, handles_all); });
}

int sba_set_fixed_points(sba_handle* h, const uint8_t* fixed_mask) {
  if (!h) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->set_fixed_points(fixed_mask); });
}

int sba_set_robust_loss(sba_handle* h, int32_t loss, double f_scale) {
  if (!h) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->set_robust_loss(loss, f_scale); });
}

int sba_time_kernel(sba_handle* h, const char* name, int32_t reps, double* mean_us_out) {
  if (!h || !name || !mean_us_out) return SBA_ERR_INVALID;
  return guarded(h, [&] { return h->eng->time_kernel(name, reps, mean_us_out); });
}

}  // extern "C"
