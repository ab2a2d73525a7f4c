// sba_engine.hpp -- the engine behind one sba_handle: device buffers, host-side layout, kernel launchers and the LM phases.
// Compiled once per camera model: SBA_NCP (11 = the reference's radial model, 13 = radial + tangential) selects the
// namespace (sba11 / sba13) all kernels live in; sba_engine_ncp11.hip / sba_engine_ncp13.hip are the two translation units.
#pragma once
#include "sba_common.hpp"
#include "sba_lm_kernels.hpp"
#include "sba_chol_blocked.hpp"
#include "sba_chol_ll.hpp"
#include "sba_chol_big.hpp"
#include "sba_sq_kernels.hpp"
#include "sba_schur_wide.hpp"
#include "sba_schur_f64.hpp"
#include "sba_ipc.hpp"

namespace SBA_NS {
using namespace sba_host;

// ============================================================================================== engine
template <typename T>
struct Engine : EngineBase {
  using T2 = typename Vec2<T>::type;
  int C = 0, N = 0;
  int64_t M = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool uploaded = false;
  bool has_w = false;
  bool identity_perm = true;
  bool dense = false;                // every point is observed by every camera exactly once
  bool dense_one_group = false;      // dense and <= 16 cameras: lane = (point, camera) kernels apply
  int nbs_dense = 1;                 // workgroups of k_backsub_dense
  // The lane = (point, camera) kernels (fused linearise + Schur, wide, dense back substitution) cost the DENSE instruction count
  // whatever the visibility; below this fraction of the N x C slots filled the observation-driven three-pass path would take over.
  // Rounds 1-3 set it to 0.35 unmeasured.  Round 4 measured it (profiles/r4_visibility_sweep*.txt: 16 / 17 cameras, 10k and 50k points,
  // fill 0.14 .. 0.45, both dtypes): the one-launch kernels win at EVERY fill a rig can have (two views of 16 cameras = 0.13: fp32
  // 78 vs 102 us per iteration at 16 x 10k, 111 vs 163 at 17 x 10k, 118 vs 155 at 16 x 50k; fp64 212 vs 245) because the three-pass
  // path is bound by its launches and pair kernels, not by the observation count.  So the threshold is 0; SBA_DENSE_MIN_VIS restores one.
  double dense_min_vis = 0.0;
  bool fused_masked = false;         // the fused kernel runs with the visibility mask
  bool lin_pts_ok = false;           // f64, one group: the point linearisation runs inside k_schur_sym (LIN)
  bool masked_ok = false;            // one group, no duplicate (point, camera) pairs, not dense: visibility mask available
  DevBuf<uint16_t> vis_mask;
  DevBuf<uint16_t> grp_mask;         // several camera groups: k_group_index tables [ngroups][N] (Schur producers)
  DevBuf<int32_t> grp_start;
  bool grp_indexed = false;
  bool pairs_fold_u = false;         // multi-group fp32: k_schur_diag_bf3 accumulates the camera blocks itself (no k_linearize_cams in the loop)
  bool no_bf3_pairs = false;         // SBA_NO_BF3_PAIRS=1: f32-input MFMA kernels for every group pair (A/B, equivalence test)
  bool no_bf3_offdiag = false;       // SBA_NO_BF3_OFFDIAG=1: ... for the off-diagonal pairs only
  bool fused_ok = false;             // dense, one camera group, f32: the linearisation runs inside the Schur kernel
  bool fused_bf3 = true;             // ... with the Schur products on the bf16 matrix pipe (k_schur_fused_bf3)
  bool fused_f64 = false;            // fp64, one group, 11 parameters: k_schur_fused_f64 (sba_schur_f64.hpp)
  int wide_ts = 1;                   // fp64 wide kernel: workgroups per slice (tile split, grid.y)
  bool fused_wide = false;           // 17 .. 23 cameras: k_schur_fused_wide (compact rows, one launch; implies fused_ok)
  int wide_pw = 2;                   // ... points per producer wave: 3 (packed lanes, 12-point rounds) for 17 and 18 cameras, else 2
  DevBuf<double> gdpart;
  std::vector<int64_t> perm;          // pm position -> caller's observation index
  int nblk = 0, nchunk = 0, ngroups = 0, npairs = 0, ksplit = 1;
  int n = 0;                          // 11*C

  // static problem data
  DevBuf<T2> uv_pm, uv_cm;
  DevBuf<T> w_pm, w_cm;
  DevBuf<int32_t> ci_pm, pi_pm, pt_start, blk_pt, pi_cm, chunk_cam, chunk_begin, chunk_end, cam_chunk_start;
  DevBuf<int32_t> pair_ga, pair_gb;
  DevBuf<int4> blk_desc;
  // parameters (double-buffered: cur / trial)
  DevBuf<double> cams[2], pts[2];
  DevBuf<T> ptsT[2], campre[2];
  int cur = 0;
  // linearization + LM work space
  DevBuf<double> V, gp, D2p, D2c, U, gc, Upart, bpart, E_own, scal_own, delta_c, cost_part, gmax_part, trial_part;
  DevBuf<T> slabs, pfac;
  DevBuf<T2> r_pm;
  DevBuf<T> Jc_pm, Jp_pm;
  // The state record lives in a two-entry buffer: k_schur_fused_bf3 can carry the accept/reject decision of the previous trial
  // step in its prologue, reading one entry and publishing the updated record to the other (see FusedDecide).  d_state.p is the
  // entry every launch enqueued NOW has to use.
  DevBuf<LMState> d_state_buf;
  struct { LMState* p = nullptr; } d_state;
  int st_slot = 0;
  bool pending_decide = false;        // lm_decide_async was called and its decision has not been enqueued yet
  const double* pend_scal = nullptr;
  int pend_ranks = 1;
  bool defer_decide = true;           // SBA_DECIDE_KERNEL=1 keeps the separate k_decide launch (A/B measurements)
  DevBuf<double> gmax_alt;            // second array of gradient maxima (the fused kernel reads one and writes the other)
  double* gmax_cur = nullptr;         // the array the last linearisation wrote
  DevBuf<sba_lm_iter_log> d_log;
  int log_read = 0;
  int cur_at_begin = 0;
  LMState* h_state = nullptr;         // pinned
  sba_lm_opts opts{};
  bool lm_active = false;
  bool chol_old = false;
  // one-workgroup factorisations of up to 256 unknowns: the right-looking all-in-LDS kernel up to 176 (k_cholesky_blocked),
  // the left-looking kernel with the factor on chip above that (k_cholesky_ll, sba_chol_ll.hpp: 17..23 cameras).
  // SBA_CHOL=ll: the left-looking kernel for every size up to 256; SBA_CHOL=blocked: never (the streamed kernel above 176) -- A/B runs, tests
  bool chol_ll = true, chol_ll_all = false;
  // fp32 engine, up to 176 unknowns: factor on f32 lanes (v_mfma_f32_16x16x4, f32 pivot chain) and repeat in f64 only when that is
  // refused -- a non-positive pivot, or one below chol_f32_tau times its diagonal entry (2^-23: the entry's own rounding noise).
  // SBA_CHOL_F32=0 keeps the f64 factorisation of rounds 1-3; SBA_CHOL_F32_TAU overrides the threshold.
  int chol_f32 = 1;
  float chol_f32_tau = 1.1920929e-7f;
  // systems larger than this take the multi-workgroup factorisation (sba_chol_big.hpp); up to CS_MAX_NB * CB = 512 unknowns the
  // streamed one-workgroup kernel could run too, but it only wins below ~210 (measured at 50k points, fp32: 17 cameras 379 vs 388 us
  // per iteration, 20: 417 vs 409, 24: 487 vs 454, 32: 647 vs 515)
  int chol_big_min_n = 209;
  DevBuf<double> chol_sol, chol_work, chol_W, chol_Minv, chol_Ld, chol_yv;
  DevBuf<int> chol_info;
  unsigned chol_epoch = 0;              // k_chol_big_back_all: launches so far (its parity picks the copy of x the launch works in)
  bool chol_big_back_one = true;        // SBA_CHOL_BIG_BACK=launches keeps one launch per block (rounds 1-3)
  DevBuf<unsigned> chol_dag_flags;      // k_chol_big_dag: Mimg_j / W(r,c) published (value = the launch's epoch)
  DevBuf<double> chol_Mimg;             // k_chol_big_dag: the factored diagonal blocks and their inverses, as they lie in LDS
  unsigned chol_dag_epoch = 0;
  bool chol_big_dag = true;             // SBA_CHOL_BIG=launches keeps one launch per block column (rounds 1-3)
  int chol_dag_max_nbr = CHOLDAG_MAX_NBR;   // SBA_CHOL_DAG_MAX_NBR: block rows up to which the one-launch factorisation is used
  bool card_shared = false;             // sba_ipc_attach found a peer rank's exchange area on THIS device (a rehearsal of N ranks on one card):
                                        // kernels whose workgroups wait for each other are then not used where a launch-per-step form exists
  bool chol_debug = false;
  bool schur_debug = false;
  int schur_exp = 0;                  // SBA_SCHUR_EXP: timing experiments of k_schur_fused_bf3 (its results are wrong when set)
  int schur_debug_skip = 0;           // SBA_SCHUR_DEBUG=k: stamps of the (k+1)-th fused launch (k >= 1: one with the decision in its prologue)
  // multi-rank (one handle per GPU, points sharded, cameras replicated): RCCL communicator + exchange buffers
  DevBuf<double> raw_uv, raw_w;         // the caller's raw arrays on the device (dense fast path of upload)
  DevBuf<long long> raw_ci, raw_pi;
  DevBuf<int> up_flag;
  DevBuf<unsigned char> pt_fixed_mask;   // sba_set_fixed_points: 1 = the point is a gauge anchor (never moves, not an unknown)
  bool has_fixed = false;
  double loss_delta = 0;                // sba_set_robust_loss: f_scale of the robust loss, 0 = linear loss
  int loss_kind = 0;                    // ... and which one (sba_loss)
  Rccl::comm_t comm = nullptr;
  int comm_rank = 0, comm_n = 1;
  DevBuf<double> xpack, sc_loc, sc_all, comm_tmp;
  double* h_comm = nullptr;             // pinned staging of the few scalars all-reduced at begin / finish
  // one-shot exchange through peer-mapped buffers instead of RCCL (sba_ipc.hpp; sba_ipc_export / sba_ipc_attach)
  bool ipc_on = false;
  bool ipc_dead = false;                // an exchange outside the LM loop timed out: the ranks' exchange counters no longer agree
  int ipc_planned = 0;                  // n_ranks the area was exported for
  double* ipc_mine = nullptr;           // this rank's area (uncached device memory)
  uint8_t ipc_my_handle[SBA_IPC_HANDLE_BYTES] = {};
  std::vector<double*> ipc_area;        // every rank's area as mapped into this process (own entry = ipc_mine)
  std::vector<char> ipc_opened;         // ... 1 where this handle called hipIpcOpenMemHandle (and has to close it)
  long long ipc_timeout_ticks = IPC_TIMEOUT_TICKS;       // SBA_IPC_TIMEOUT_S: how long a gate waits for a peer (100 MHz ticks)
  DevBuf<double*> ipc_ptrs;             // ... the same table on the device
  DevBuf<int> ipc_fail;                 // set by a gate that ran out of time outside the LM loop
  IpcLayout ipc_L{};
  unsigned long long ipc_seq[IPC_KINDS] = {0, 0, 0};     // exchanges enqueued so far, per kind (the value the flags carry)
  bool multi() const { return comm != nullptr || ipc_on; }
  long long N_global = 0;
  DevBuf<long long> schur_dbg;
  DevBuf<long long> chol_dbg;
  double initial_cost = 0;
  std::vector<sba_lm_iter_log> log;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_wait = nullptr;
  // in-loop kernel timing (opts.reserved[0] != 0): one HIP event pair per kernel class per LM iteration
  enum { KP_LINP = 0, KP_LINC, KP_SCHUR, KP_REDUCE, KP_CHOL, KP_BACKSUB, KP_N };
  hipEvent_t pev[KP_N][2] = {};
  bool pev_used[KP_N] = {};
  bool prof_on = false;
  double prof_us[KP_N] = {};
  long long prof_cnt[KP_N] = {};
  int pslot = 0;
  void pslot_advance() {}
  void prof_begin(int k) { if (prof_on) HIPCHK(hipEventRecord(pev[k][0], stream)); }
  void prof_end(int k) { if (prof_on) { HIPCHK(hipEventRecord(pev[k][1], stream)); pev_used[k] = true; } }
  void prof_collect() {   // call after a stream sync
    if (!prof_on) return;
    for (int k = 0; k < KP_N; ++k)
      if (pev_used[k]) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pev[k][0], pev[k][1]) == hipSuccess) { prof_us[k] += (double)ms * 1e3; prof_cnt[k]++; }
        pev_used[k] = false;
      }
  }

  HostRes hres;
  bool have_hres = false;
  // small device-to-host results land in pinned memory (a copy into pageable memory is staged and blocks the caller per copy)
  double* h_land = nullptr;
  static constexpr size_t LAND_DOUBLES = (HostRes::PINNED_BYTES - HostRes::PINNED_STATE_BYTES) / sizeof(double);
  bool poll_clean = false;            // nothing has been enqueued since the last lm_poll: its state and log are current
  int n_decides = 0;                  // accept/reject kernels enqueued since lm_begin (upper bound of the log rows on the device)
  ~Engine() override {
    if (comm) (void)Rccl::get().comm_destroy(comm);
    if (stream) (void)hipStreamSynchronize(stream);
    ipc_close_peers();
    if (ipc_mine) { IpcLocalAreas::get().remove(ipc_my_handle); (void)hipFree(ipc_mine); }
    if (h_comm) (void)hipHostFree(h_comm);
    if (have_hres) {
      if (stream) (void)hipStreamSynchronize(stream);     // nothing of this handle may still be in flight when its buffers are recycled
      (void)hipStreamSynchronize(hres.stream);
      if (hres.copy_stream) (void)hipStreamSynchronize(hres.copy_stream);
      HostResPool::get().give(hres);
    }
  }

  void init(const sba_problem_desc& d) override {
    C = d.n_cams; N = d.n_points; M = d.n_obs; device = d.device; n = C * NCP;
    arena.device = device;
    HIPCHK(hipSetDevice(device));
    static_assert(sizeof(LMState) <= 1024 && KP_N * 2 + 2 <= (int)(sizeof(HostRes::ev) / sizeof(hipEvent_t)), "HostRes sizes");
    if (!HostResPool::get().take(device, &hres)) {
      hres.device = device;
      HIPCHK(hipStreamCreateWithFlags(&hres.stream, hipStreamNonBlocking));
      HIPCHK(hipStreamCreateWithFlags(&hres.copy_stream, hipStreamNonBlocking));
      HIPCHK(hipHostMalloc(&hres.pinned, HostRes::PINNED_BYTES, hipHostMallocDefault));
      for (auto& e : hres.ev) HIPCHK(hipEventCreate(&e));
      HIPCHK(hipEventCreateWithFlags(&hres.ev_wait, hipEventDisableTiming));
    }
    have_hres = true;
    if (d.use_stream) { stream = reinterpret_cast<hipStream_t>(d.stream); own_stream = false; }
    else { stream = hres.stream; own_stream = true; }
    h_state = static_cast<LMState*>(hres.pinned);
    h_land = reinterpret_cast<double*>(static_cast<char*>(hres.pinned) + HostRes::PINNED_STATE_BYTES);
    ev0 = hres.ev[0]; ev1 = hres.ev[1]; ev_wait = hres.ev_wait;
    for (int k = 0; k < KP_N; ++k) for (int j = 0; j < 2; ++j) pev[k][j] = hres.ev[2 + 2 * k + j];
    d_state_buf.alloc(2);
    d_state.p = d_state_buf.p;
    if (getenv("SBA_DECIDE_KERNEL")) defer_decide = false;
    if (const char* e = getenv("SBA_DENSE_MIN_VIS")) {
      char* end = nullptr;
      const double v = strtod(e, &end);
      if (end != e && v >= 0 && v <= 1) dense_min_vis = v;
    }
    // SBA_FUSED_MFMA=f32 keeps the f32-input MFMA kernel (A/B measurements, equivalence test); default: bf16 x 3 split
    if (const char* e = getenv("SBA_FUSED_MFMA")) fused_bf3 = (std::string(e) != "f32");
    if (const char* e = getenv("SBA_CHOL")) {
      chol_old = (std::string(e) == "old"); chol_ll = (std::string(e) != "blocked") && !chol_old; chol_ll_all = (std::string(e) == "ll");
    }
    if (getenv("SBA_CHOL_DEBUG")) chol_debug = true;
    if (const char* e = getenv("SBA_SCHUR_EXP")) schur_exp = atoi(e);
    if (const char* e = getenv("SBA_CHOL_BIG_BACK")) chol_big_back_one = std::string(e) != "launches";
    if (const char* e = getenv("SBA_CHOL_BIG")) chol_big_dag = std::string(e) != "launches";
    if (const char* e = getenv("SBA_LINP_PF")) linp_pf = atoi(e) != 0;
    if (const char* e = getenv("SBA_CHOL_DAG_MAX_NBR")) chol_dag_max_nbr = std::max(1, std::min(CHOLDAG_MAX_NBR, atoi(e)));
    if (const char* e = getenv("SBA_CHOL_F32")) chol_f32 = atoi(e) != 0;
    if (const char* e = getenv("SBA_CHOL_F32_TAU")) { char* end = nullptr; const double v = strtod(e, &end); if (end != e && v >= 0 && v < 1) chol_f32_tau = (float)v; }
    if (const char* e = getenv("SBA_CHOL_BIG_MIN_N")) {      // diagnostic: route smaller systems through the big path too
      char* end = nullptr;
      const long v = strtol(e, &end, 10);
      if (end != e && *end == 0 && v >= 0) chol_big_min_n = (int)std::min<long>(v, CS_MAX_NB * CB);
    }
    if (const char* e = getenv("SBA_SCHUR_DEBUG")) { schur_debug = true; schur_debug_skip = std::max(0, atoi(e) - 1); schur_dbg.alloc(64); schur_dbg.zero(stream); }
    static bool attrs_set[16] = {};          // the function attributes are per process (and device), not per handle
    if (device < 16 && attrs_set[device]) return;
    // kernels whose dynamic LDS can exceed the 64 KB default
    // only the Schur flavours this dtype launches are instantiated (f32: producer/consumer + fused; f64: symmetric + PARTIAL)
    auto big_lds = [&](const void* fn) { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024); };
    if constexpr (SCHUR_SYM<T>) {
      HIPCHK(big_lds(reinterpret_cast<const void*>(&k_schur_sym<T, true, false>)));
      HIPCHK(big_lds(reinterpret_cast<const void*>(&k_schur_sym<T, false, false>)));
      HIPCHK(big_lds(reinterpret_cast<const void*>(&k_schur_sym<T, true, true>)));
      HIPCHK(big_lds(reinterpret_cast<const void*>(&k_schur_sym<T, false, true>)));
      if constexpr (SCHUR_LIN_OK<T>) HIPCHK(big_lds(reinterpret_cast<const void*>(&k_schur_sym<T, true, false, true>)));
#if SBA_NCP == 11
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_f64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurF64Cfg::LDS_BYTES));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_wide_f64<12, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurWide64Cfg<12, 2, 1>::LDS_BYTES));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_wide_f64<13, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurWide64Cfg<13, 2, 1>::LDS_BYTES));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_wide_f64<12, 3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurWide64Cfg<12, 3, 1>::LDS_BYTES));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_wide_f64<13, 3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurWide64Cfg<13, 3, 1>::LDS_BYTES));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_wide_f64<14, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurWide64Cfg<14, 2, 2>::LDS_BYTES));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_wide_f64<15, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurWide64Cfg<15, 2, 2>::LDS_BYTES));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_wide_f64<16, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurWide64Cfg<16, 2, 2>::LDS_BYTES));
#endif
    } else {
      HIPCHK(big_lds(reinterpret_cast<const void*>(&k_schur<T, true, false>)));
      HIPCHK(big_lds(reinterpret_cast<const void*>(&k_schur<T, false, false>)));
#if SBA_NCP == 11
      HIPCHK(big_lds(reinterpret_cast<const void*>(&k_schur_fused)));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_bf3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurBf3Cfg::LDS_BYTES));
#endif
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_wide<12, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurWideCfg<12, 2>::LDS_BYTES));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_wide<13, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurWideCfg<13, 2>::LDS_BYTES));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_wide<14, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurWideCfg<14, 2>::LDS_BYTES));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_wide<15, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurWideCfg<15, 2>::LDS_BYTES));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_wide<16, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurWideCfg<16, 2>::LDS_BYTES));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_wide<12, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurWideCfg<12, 3>::LDS_BYTES));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_wide<13, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurWideCfg<13, 3>::LDS_BYTES));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_wide<8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurWideCfg<8, 2>::LDS_BYTES));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_fused_wide<8, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurWideCfg<8, 3>::LDS_BYTES));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_diag_bf3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurPairCfg::LDS_BYTES));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_offdiag_bf3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SchurBf3OffCfg::LDS_BYTES));
    }
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_resjac<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cholesky_solve<true, T>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    // (dynamic: the block triangle of 176 unknowns in doubles + two vectors = 146,432 B; the kernel's static LDS -- tables, damping,
    //  right-hand side, trial cameras -- comes on top of it and both must fit the 160 KB of a CU)
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cholesky_blocked<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)((CHOLB_MAX_NB * (CHOLB_MAX_NB + 1) / 2 * CBS + 2 * CHOLB_MAX_NB * CB) * sizeof(double))));
    if constexpr (sizeof(T) == 4) {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cholesky_blocked<T, 16, 20>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)((14 * 15 / 2 * CB * 20 + 2 * 14 * CB) * sizeof(float))));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cholesky_blocked<T, 16, 17>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)((16 * 17 / 2 * CB * 17 + 2 * 16 * CB) * sizeof(float))));
    }
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cholesky_stream), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cholesky_ll<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CLL_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_big_step), hipFuncAttributeMaxDynamicSharedMemorySize, CHOLBIG_LDS_BLOCKS * CBS * (int)sizeof(double)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_big_dag<double>), hipFuncAttributeMaxDynamicSharedMemorySize, CHOLBIG_LDS_BLOCKS * CBS * (int)sizeof(double)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_big_dag<float>), hipFuncAttributeMaxDynamicSharedMemorySize, CHOLBIG_LDS_BLOCKS * CholLay<float>::BS * (int)sizeof(float)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_backsub_trial<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linearize_points<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_residual<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sq_linearize<T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sq_linearize<T, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sq_trial<T, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sq_trial<T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    if (device < 16) attrs_set[device] = true;
  }

  void sync() { HIPCHK(hipStreamSynchronize(stream)); }
  // Wait for the stream with a short busy poll first: a blocking hipStreamSynchronize wakes the thread tens of microseconds after the
  // GPU is done, which is a visible share of a 20-iteration solve (2.6 ms); after SPIN_US the thread blocks like sync() does.
  void sync_spin() {
    constexpr double SPIN_US = 5000.0;
    HIPCHK(hipEventRecord(ev_wait, stream));
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t q = hipEventQuery(ev_wait);
      if (q == hipSuccess) return;
      if (q != hipErrorNotReady) { HIPCHK(q); }
      if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > SPIN_US) break;
    }
    HIPCHK(hipStreamSynchronize(stream));
  }

  // ------------------------------------------------------------------ upload + host-side layout
  int upload(const double* cams_h, const double* pts_h, const double* uv_h, const int64_t* ci_h,
             const int64_t* pi_h, const double* w_h) override {
    HIPCHK(hipSetDevice(device));
    const bool up_dbg = getenv("SBA_UPLOAD_DEBUG") != nullptr;
    auto up_t0 = std::chrono::steady_clock::now();
    auto up_lap = [&](const char* what) {
      if (!up_dbg) return;
      const auto t = std::chrono::steady_clock::now();
      fprintf(stderr, "[upload] %-28s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(t - up_t0).count());
      up_t0 = t;
    };
    if (C <= 0 || N < 0 || M < 0) { err = "bad problem size"; return SBA_ERR_INVALID; }
    if (C > 128) { err = "more than 128 cameras is not supported yet"; return SBA_ERR_UNSUPPORTED; }
    if (M > (int64_t)0x7fffffff - 1024) { err = "too many observations for int32 device indices"; return SBA_ERR_UNSUPPORTED; }
    has_w = (w_h != nullptr);
    // Device-side layout: when the list can only be the canonical dense one -- M = N*C -- the caller's raw
    // float64 / int64 arrays go to the GPU as they are and ONE kernel validates them (observation i must be point i / C, camera
    // i % C: that single test implies in-range, point-major, camera-minor, no duplicates, every camera sees every point),
    // narrows them to the device types and writes the point-major and the camera-major copies.  Anything else (sparse
    // visibility, unsorted or duplicated observations) takes the host path below.
    bool dev_dense = false;
    std::vector<int32_t> ptstart;
    bool sorted = true, cam_sorted = true;
    std::vector<uint16_t> vmask;
    bool nodup = (C <= GROUP_CAMS);
    if (M > 0 && M == (int64_t)N * C && !getenv("SBA_HOST_LAYOUT")) {
      raw_uv.alloc((size_t)M * 2); raw_ci.alloc(M); raw_pi.alloc(M);
      if (has_w) raw_w.alloc(M);
      uv_pm.alloc(M); ci_pm.alloc(M); pi_pm.alloc(M); uv_cm.alloc(M); pi_cm.alloc(M); pt_start.alloc((size_t)N + 1);
      if (has_w) { w_pm.alloc(M); w_cm.alloc(M); }
      if (up_flag.n == 0) up_flag.alloc(1);
      up_flag.zero(stream);
      HIPCHK(hipMemcpyAsync(raw_uv.p, uv_h, sizeof(double) * 2 * M, hipMemcpyHostToDevice, stream));
      HIPCHK(hipMemcpyAsync(raw_ci.p, ci_h, sizeof(int64_t) * M, hipMemcpyHostToDevice, stream));
      HIPCHK(hipMemcpyAsync(raw_pi.p, pi_h, sizeof(int64_t) * M, hipMemcpyHostToDevice, stream));
      if (has_w) HIPCHK(hipMemcpyAsync(raw_w.p, w_h, sizeof(double) * M, hipMemcpyHostToDevice, stream));
      up_lap("raw H2D");
      hipLaunchKernelGGL(k_upload_dense<T>, dim3((unsigned)((std::max<int64_t>(M, N + 1) + 255) / 256)), dim3(256), 0, stream,
                         reinterpret_cast<const double2*>(raw_uv.p), raw_ci.p, raw_pi.p, has_w ? raw_w.p : (const double*)nullptr, C, N, (long long)M,
                         uv_pm.p, ci_pm.p, pi_pm.p, has_w ? w_pm.p : (T*)nullptr, uv_cm.p, pi_cm.p, has_w ? w_cm.p : (T*)nullptr,
                         pt_start.p, up_flag.p);
      int flag = 1;
      HIPCHK(hipMemcpyAsync(&flag, up_flag.p, sizeof(int), hipMemcpyDeviceToHost, stream));
      sync();
      up_lap("device validate + layout");
      dev_dense = (flag == 0);
    }
    if (dev_dense) {
      ptstart.resize((size_t)N + 1);
      for (int p = 0; p <= N; ++p) ptstart[p] = p * C;
      dense = true; identity_perm = true; perm.clear();
      if (C > PM_BLOCK) { err = "a point has more than 256 observations"; return SBA_ERR_UNSUPPORTED; }
    } else {
    // one pass: range check, point-major order, and camera order inside a point (strictly increasing cameras inside every
    // point = no duplicate (point, camera) pair and already canonical: what get_points3d.py:78-86 emits)
    {
      int64_t bad[4] = {-1, -1, -1, -1};
      bool uns[4] = {false, false, false, false}, cuns[4] = {false, false, false, false};
      par_for(M, [&](int64_t lo, int64_t hi, int t) {
        for (int64_t i = lo; i < hi; ++i) {
          if (ci_h[i] < 0 || ci_h[i] >= C || pi_h[i] < 0 || pi_h[i] >= N) { if (bad[t] < 0) bad[t] = i; continue; }
          if (i) {
            if (pi_h[i] < pi_h[i - 1]) uns[t] = true;
            else if (pi_h[i] == pi_h[i - 1] && ci_h[i] <= ci_h[i - 1]) cuns[t] = true;
          }
        }
      });
      for (int t = 0; t < 4; ++t) {
        if (bad[t] >= 0) { err = "camera/point index out of range at observation " + std::to_string(bad[t]); return SBA_ERR_INVALID; }
        sorted = sorted && !uns[t];
        cam_sorted = cam_sorted && !cuns[t];
      }
    }
    // point-major order (stable counting sort by point)
    ptstart.assign((size_t)N + 1, 0);
    for (int64_t i = 0; i < M; ++i) ptstart[pi_h[i] + 1]++;
    int maxdeg = 0;
    for (int p = 0; p < N; ++p) { maxdeg = std::max(maxdeg, ptstart[p + 1]); ptstart[p + 1] += ptstart[p]; }
    dense = (M == (int64_t)N * C);
    for (int p = 0; p < N && dense; ++p) dense = (ptstart[p + 1] - ptstart[p] == C);
    if (maxdeg > PM_BLOCK) { err = "a point has more than 256 observations"; return SBA_ERR_UNSUPPORTED; }
    perm.resize(M);
    identity_perm = sorted;
    if (sorted) par_for(M, [&](int64_t lo, int64_t hi, int) { for (int64_t i = lo; i < hi; ++i) perm[i] = i; });   // (the canonical pass below may still reorder)
    else {
      std::vector<int32_t> fill(ptstart.begin(), ptstart.end() - 1);
      for (int64_t i = 0; i < M; ++i) perm[fill[pi_h[i]]++] = i;
    }
    // One camera group: put the observations of every point in camera order (what get_points3d.py:78-86 emits anyway) and
    // record which cameras see it.  Without duplicate (point, camera) pairs the lane = (point, camera) kernels apply:
    // dense = every camera sees every point (observation (p, c) at p*C + c), else through the visibility mask.
    if (nodup && sorted && cam_sorted) {           // already canonical: only the visibility masks are needed
      vmask.assign(N, 0);
      for (int64_t i = 0; i < M; ++i) vmask[pi_h[i]] |= (uint16_t)(1u << ci_h[i]);
    } else if (nodup) {
      vmask.assign(N, 0);
      std::vector<int64_t> slot(C);
      for (int p = 0; p < N && nodup; ++p) {
        std::fill(slot.begin(), slot.end(), (int64_t)-1);
        const int a = ptstart[p], b = ptstart[p + 1];
        for (int k = a; k < b; ++k) {
          const int64_t i = perm[k];
          if (slot[ci_h[i]] >= 0) { nodup = false; break; }
          slot[ci_h[i]] = i;
          vmask[p] |= (uint16_t)(1u << ci_h[i]);
        }
        if (!nodup) break;
        int k = a;
        for (int c = 0; c < C; ++c)
          if (slot[c] >= 0) { if (perm[k] != slot[c]) { perm[k] = slot[c]; identity_perm = false; } ++k; }
      }
    }
    dense = dense && nodup;
    up_lap("validate + sort + canonical");
    }   // host validation
    masked_ok = nodup && !dense;
    std::vector<T2> uvp(dev_dense ? 0 : M);
    std::vector<T> wp(has_w && !dev_dense ? M : 0);
    std::vector<int32_t> cip(dev_dense ? 0 : M), pip(dev_dense ? 0 : M);
    if (!dev_dense) par_for(M, [&](int64_t lo, int64_t hi, int) {
      for (int64_t k = lo; k < hi; ++k) {
        const int64_t i = perm[k];
        uvp[k].x = (T)uv_h[2 * i]; uvp[k].y = (T)uv_h[2 * i + 1];
        if (has_w) wp[k] = (T)w_h[i];
        cip[k] = (int32_t)ci_h[i]; pip[k] = (int32_t)pi_h[i];
      }
    });
    up_lap("permute observations");
    // point-aligned blocks of <= 256 observations
    std::vector<int32_t> blk;
    blk.push_back(0);
    {
      int p = 0;
      while (p < N) {
        int q = p; int cnt = 0;
        while (q < N && cnt + (ptstart[q + 1] - ptstart[q]) <= PM_BLOCK && (q - p) < PM_BLOCK) { cnt += ptstart[q + 1] - ptstart[q]; ++q; }
        if (q == p) ++q;   // cannot happen (maxdeg <= 256) but never loop forever
        blk.push_back(q);
        p = q;
      }
    }
    nblk = (int)blk.size() - 1;
    std::vector<int4> bdesc(nblk);
    for (int b = 0; b < nblk; ++b) bdesc[b] = make_int4(blk[b], blk[b + 1], ptstart[blk[b]], ptstart[blk[b + 1]]);
    // camera-major order of the pm list (stable => points ascending inside a camera)
    std::vector<int32_t> camcount(C + 1, 0);
    if (dev_dense) { for (int c = 0; c <= C; ++c) camcount[c] = c * N; }
    else {
      for (int64_t k = 0; k < M; ++k) camcount[cip[k] + 1]++;
      for (int c = 0; c < C; ++c) camcount[c + 1] += camcount[c];
    }
    std::vector<T2> uvc(dev_dense ? 0 : M);
    std::vector<T> wc(has_w && !dev_dense ? M : 0);
    std::vector<int32_t> pic(dev_dense ? 0 : M);
    if (!dev_dense) {
      std::vector<int32_t> fill(camcount.begin(), camcount.end() - 1);
      for (int64_t k = 0; k < M; ++k) {
        const int32_t d = fill[cip[k]]++;
        uvc[d] = uvp[k]; pic[d] = pip[k];
        if (has_w) wc[d] = wp[k];
      }
    }
    std::vector<int32_t> ch_cam, ch_beg, ch_end, cam_ch(C + 1, 0);
    for (int c = 0; c < C; ++c) {
      cam_ch[c] = (int32_t)ch_cam.size();
      for (int32_t b = camcount[c]; b < camcount[c + 1]; b += CM_CHUNK) {
        ch_cam.push_back(c); ch_beg.push_back(b); ch_end.push_back(std::min(camcount[c + 1], b + CM_CHUNK));
      }
    }
    cam_ch[C] = (int32_t)ch_cam.size();
    nchunk = (int)ch_cam.size();
    // camera groups / pairs for the Schur kernel
    ngroups = (C + GROUP_CAMS - 1) / GROUP_CAMS;
    std::vector<int32_t> pga, pgb;
    for (int a = 0; a < ngroups; ++a) { pga.push_back(a); pgb.push_back(a); }   // diagonal pairs first
    for (int a = 0; a < ngroups; ++a)
      for (int b = a + 1; b < ngroups; ++b) { pga.push_back(a); pgb.push_back(b); }
    npairs = (int)pga.size();
    {
      int target = 256;
      if (const char* e = getenv("SBA_SCHUR_WGS")) target = std::max(1, atoi(e));
      // workgroups per k-split: every pair is dealt to TS workgroups (tile split, grid.z of k_schur)
      // the diagonal and the off-diagonal pairs are two launches, one after the other, with the same k-split: it is sized so
      // that the SMALLER of the two still fills the chip (sized for the larger one, the 4 diagonal pairs of a 64-camera rig ran
      // on 84 of 256 CUs); the larger launch then simply takes several rounds of shorter workgroups
      const int wg_diag = ngroups * SchurSel<T, true>::TS, wg_off = (npairs - ngroups) * SchurSel<T, false>::TS;
      const int wg_per_ks = wg_off > 0 ? std::min(wg_diag, wg_off) : wg_diag;
      int ks = std::max(1, (target + wg_per_ks - 1) / wg_per_ks);
      const int maxks = std::max(1, (N + SCHUR_PTS - 1) / SCHUR_PTS);
      ksplit = std::min(ks, maxks);
      // Round 4: several camera groups -- the launches run in ROUNDS of one workgroup per CU, and a last round that is half empty costs
      // as much as a full one (64 cameras, k-split 64: 384 off-diagonal workgroups = 1.5 rounds).  Measured (profiles/r4_ksplit_sweep.txt):
      // 64 x 200k 3 674 -> 3 330 us per iteration with twice the k-split (512 + 768 workgroups: whole rounds, and a finer tail), 128 x 125k
      // with 13 parameters 10.67 -> 9.98 ms, 32 x 50k 470 -> 435 us.  So: aim at two rounds for the smaller launch, then pick, in a window
      // around that, the k-split with the smallest modelled makespan  sum over the two launches of  rounds x points per workgroup
      // (an off-diagonal workgroup builds two panels and 121 tiles: weight 2).  SBA_SCHUR_WGS pins the target and skips the search.
      if (ngroups > 1 && !getenv("SBA_SCHUR_WGS")) {
        int ncu = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
        const int ks0 = std::max(1, (2 * ncu + wg_per_ks - 1) / wg_per_ks);
        double best = 1e300;
        int best_ks = ks0;
        for (int k = std::max(1, ks0 * 3 / 4); k <= ks0 * 3 / 2; ++k) {
          const double per = std::ceil((double)N / k);
          const double cost = std::ceil((double)wg_diag * k / ncu) * per + (wg_off > 0 ? 2.0 * std::ceil((double)wg_off * k / ncu) * per : 0.0);
          if (cost < best * (1 - 1e-9)) { best = cost; best_ks = k; }
        }
        ksplit = std::min(best_ks, maxks);
      }
    }

    up_lap("blocks + camera-major copies");
    if (!dev_dense) {
      uv_pm.upload(uvp, stream); ci_pm.upload(cip, stream); pi_pm.upload(pip, stream);
      if (has_w) { w_pm.upload(wp, stream); w_cm.upload(wc, stream); }
      pt_start.upload(ptstart, stream);
      uv_cm.upload(uvc, stream); pi_cm.upload(pic, stream);
    }
    blk_pt.upload(blk, stream); blk_desc.upload(bdesc, stream);
    chunk_cam.upload(ch_cam, stream); chunk_begin.upload(ch_beg, stream); chunk_end.upload(ch_end, stream);
    cam_chunk_start.upload(cam_ch, stream);
    pair_ga.upload(pga, stream); pair_gb.upload(pgb, stream);
    grp_indexed = false;
    no_bf3_pairs = getenv("SBA_NO_BF3_PAIRS") != nullptr;
    no_bf3_offdiag = getenv("SBA_NO_BF3_OFFDIAG") != nullptr;
    if (ngroups > 1 && N > 0 && !getenv("SBA_SCHUR_SCAN")) {
      grp_mask.alloc((size_t)ngroups * N); grp_start.alloc((size_t)ngroups * N);
      if (up_flag.n == 0) up_flag.alloc(1);
      up_flag.zero(stream);
      hipLaunchKernelGGL(k_group_index, dim3((N + 255) / 256), dim3(256), 0, stream, ci_pm.p, pt_start.p, N, ngroups, grp_mask.p,
                         grp_start.p, up_flag.p);
      int flag = 1;
      HIPCHK(hipMemcpyAsync(&flag, up_flag.p, sizeof(int), hipMemcpyDeviceToHost, stream));
      sync();
      grp_indexed = (flag == 0);     // cameras strictly ascending inside every point; otherwise the producers scan
    }
    // Up to 256 reduced-system rows, f32: the one-launch fused kernel of sba_schur_wide.hpp -- 17 .. 23 cameras of the 11-parameter
    // model (one group has k_schur_fused_bf3), every rig of up to 19 cameras of the 13-parameter model.  Like the masked
    // one-group kernel its producer cost does not shrink with the visibility, so below ~35 % the three-pass path stays.
    fused_wide = false;
    if constexpr (sizeof(T) == 4) {
      const bool dense_enough = (double)M >= dense_min_vis * (double)N * C;
      const bool rig_ok = C > GROUP_CAMS ? (dense || (grp_indexed && dense_enough))
                                         : (NCP != 11 && !getenv("SBA_NO_DENSE") && (dense || (masked_ok && dense_enough)));
      fused_wide = rig_ok && C * NCP <= 16 * WIDE_MAX_NTW && N > 0 && fused_bf3 && !getenv("SBA_NO_FUSED") && !getenv("SBA_NO_WIDE");
      if (fused_wide) {
        int target = 256;
        if (const char* e = getenv("SBA_SCHUR_WGS")) target = std::max(1, atoi(e));
        wide_pw = (wide_ntw(C) <= 13 && 3 * C <= 64 && !getenv("SBA_WIDE_PW2")) ? 3 : 2;
        ksplit = std::max(1, std::min(target, (N + 4 * wide_pw - 1) / (4 * wide_pw)));
      }
    }
#if SBA_NCP == 11
    // fp64, 17 and 18 cameras (12 / 13 row tiles): k_schur_fused_wide_f64 (sba_schur_f64.hpp), same eligibility; SBA_NO_FUSED64=1 or
    // SBA_NO_WIDE=1 keep the pair kernels
    if constexpr (sizeof(T) == 8) {
      const bool dense_enough = (double)M >= dense_min_vis * (double)N * C;
      fused_wide = C > GROUP_CAMS && C * NCP <= 16 * WIDE_MAX_NTW && (dense || (grp_indexed && dense_enough)) && N > 0 &&
                   !getenv("SBA_NO_FUSED") && !getenv("SBA_NO_FUSED64") && !getenv("SBA_NO_WIDE");
      if (fused_wide) {
        int target = 256;
        if (const char* e = getenv("SBA_SCHUR_WGS")) target = std::max(1, atoi(e));
        // 17, 18 cameras (12, 13 row tiles): one workgroup per slice, three points per producer wave; 19 .. 23 (14 .. 16 tiles): two
        // workgroups per slice share its tiles (a consumer wave holds at most ~23 f64 accumulator tiles), two points per wave
        wide_ts = wide_ntw(C) > 13 ? 2 : 1;
        wide_pw = (wide_ts == 1 && !getenv("SBA_WIDE_PW2")) ? 3 : 2;
        ksplit = std::max(1, std::min(target / wide_ts, (N + 4 * wide_pw - 1) / (4 * wide_pw)));
      }
    }
#endif

    for (int b = 0; b < 2; ++b) {
      cams[b].alloc((size_t)C * NCP); pts[b].alloc((size_t)N * 3);
      ptsT[b].alloc((size_t)N * 3); campre[b].alloc((size_t)C * CAMPRE);
    }
    pfac.alloc((size_t)std::max(N, 1) * PF);
    V.alloc((size_t)N * 6); gp.alloc((size_t)N * 3); D2p.alloc((size_t)N * 3); D2c.alloc(n);
    U.alloc((size_t)C * NCP * NCP); gc.alloc(n); Upart.alloc((size_t)std::max(1, nchunk) * 256);
    bpart.alloc(std::max((size_t)ngroups * ksplit * GROUP_ROWS, fused_wide ? (size_t)ksplit * WIDE_ROWS : (size_t)0));
    slabs.alloc(std::max((size_t)npairs * ksplit * GROUP_TILES * GROUP_TILES * 256, fused_wide ? (size_t)ksplit * WIDE_SLOTS * 256 : (size_t)0));
    E_own.alloc((size_t)n * n + 3 * n + 1); scal_own.alloc(NSCAL); delta_c.alloc(n);
    const int nres_blocks = (int)((M + PM_BLOCK - 1) / PM_BLOCK);
    dense_one_group = dense && C <= GROUP_CAMS && N > 0 && !getenv("SBA_NO_DENSE");
    // 17 .. 23 cameras, dense or group-indexed and dense enough (the rigs of the wide fused kernels): the same back substitution
    // with a point per 32-lane wave half
    backsub_wide = C > GROUP_CAMS && C <= 32 && N > 0 && (dense || (grp_indexed && (double)M >= dense_min_vis * (double)N * C)) &&
                   !getenv("SBA_NO_DENSE") && !getenv("SBA_NO_WIDE");
    backsub_pack = backsub_wide && 3 * C <= 64 && !getenv("SBA_WIDE_PW2");       // three points per wave (17 .. 21 cameras)
    // 33 .. 128 cameras (round 4): a point per wave, up to two cameras per lane (k_backsub_dense<T, 64>)
    backsub_wave = C > 32 && N > 0 && (dense || (grp_indexed && (double)M >= dense_min_vis * (double)N * C)) && !getenv("SBA_NO_DENSE") && !getenv("SBA_NO_WIDE");
    const int bs_ppc = backsub_wave ? 4 : backsub_pack ? 12 : backsub_wide ? 8 : 16;
    linp_lw = getenv("SBA_LINP_BLOCKS") ? 0 : backsub_wave ? 64 : (backsub_wide && C > 23) ? 32 : 0;
    nlinp = linp_lw ? std::max(1, std::min((N + (PM_BLOCK / linp_lw) - 1) / (PM_BLOCK / linp_lw), 1024)) : 0;
    nbs_dense = std::max(1, std::min((N + bs_ppc - 1) / bs_ppc, getenv("SBA_BS_WGS") ? atoi(getenv("SBA_BS_WGS")) : (sizeof(T) == 4 ? 768 : 512)));
    // the fused linearise+Schur kernel also serves sparse one-group rigs through the visibility mask; its producer cost
    // does not shrink with the number of observations, so below ~35 % visibility the three-pass path is used
    const bool masked_fused = masked_ok && N > 0 && (double)M >= dense_min_vis * (double)N * C && !getenv("SBA_NO_DENSE");
    // (the fused kernel is built for the 11-parameter model only: 77 register accumulators per lane; 13 parameters need 104)
    // (fp64, round 3: k_schur_fused_f64, same structure on the f64 matrix pipe; SBA_NO_FUSED64=1 keeps the three-launch path)
    fused_f64 = NCP == 11 && sizeof(T) == 8 && (dense_one_group || masked_fused) && !getenv("SBA_NO_FUSED") && !getenv("SBA_NO_FUSED64");
    fused_ok = (NCP == 11 && (dense_one_group || masked_fused) && sizeof(T) == 4 && !getenv("SBA_NO_FUSED")) || fused_wide || fused_f64;
    fused_masked = fused_ok && !dense_one_group && !fused_wide;
    lin_pts_ok = (dense_one_group || masked_fused) && SCHUR_LIN_OK<T> && !getenv("SBA_NO_FUSED") && !fused_f64;
    // (the lane = (point, camera) back substitution serves sparse one-group rigs through the same mask, down to the visibility
    //  where the point-aligned kernel, whose cost follows the observation count, wins)
    backsub_masked = masked_ok && C <= GROUP_CAMS && N > 0 && (double)M >= dense_min_vis * (double)N * C && !getenv("SBA_NO_DENSE");
    if (fused_masked || (lin_pts_ok && !dense_one_group) || (fused_wide && C <= GROUP_CAMS && !dense) || backsub_masked) vis_mask.upload(vmask, stream);
    if (fused_ok) gdpart.alloc((size_t)ksplit * 2 * (fused_wide ? WIDE_ROWS : GROUP_ROWS));
    // several camera groups on the bf16 pair kernels (round 4): the diagonal pairs also accumulate U_c / g_c (SBA_PAIRS_LINC=1 keeps
    // k_linearize_cams + k_reduce_cams)
    pairs_fold_u = sizeof(T) == 4 && ngroups > 1 && grp_indexed && !no_bf3_pairs && !fused_ok && !getenv("SBA_PAIRS_LINC");
    if (pairs_fold_u) gdpart.alloc((size_t)ngroups * ksplit * 2 * GROUP_ROWS);
    cost_part.alloc((size_t)std::max(std::max(std::max(std::max(nblk, nres_blocks), ksplit), nlinp), 1)); gmax_part.alloc(std::max(std::max(std::max(nblk, ksplit), nlinp), 1)); gmax_alt.alloc(std::max(std::max(std::max(nblk, ksplit), nlinp), 1));
    // (k_backsub_dense writes one partial row per WORKGROUP: nbs_dense of them, which exceeds the number of point-aligned blocks on
    //  dense rigs with few cameras -- 8 x 1000: 63 vs 32; sized for nblk alone the rows used to run over into the next buffer)
    trial_part.alloc((size_t)4 * std::max(std::max(nblk, nbs_dense), 1));
    up_lap("allocations + H2D enqueue");
    sync();   // the staging vectors go out of scope now
    up_lap("H2D completion");
    uploaded = true;
    cur = 0;
    set_params(cams_h, pts_h);
    return SBA_OK;
  }

  void set_params(const double* cams_h, const double* pts_h) {
    HIPCHK(hipMemcpyAsync(cams[cur].p, cams_h, sizeof(double) * C * NCP, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(pts[cur].p, pts_h, sizeof(double) * (size_t)N * 3, hipMemcpyHostToDevice, stream));
    std::vector<T> pt((size_t)N * 3);
    for (size_t i = 0; i < pt.size(); ++i) pt[i] = (T)pts_h[i];
    HIPCHK(hipMemcpyAsync(ptsT[cur].p, pt.data(), sizeof(T) * pt.size(), hipMemcpyHostToDevice, stream));
    cam_prep(cur);
    sync();
  }

  void cam_prep(int b) {
    hipLaunchKernelGGL(k_cam_prep<T>, dim3((C + 63) / 64), dim3(64), 0, stream, cams[b].p, campre[b].p, C);
  }

  size_t lds_cams() const { return (size_t)C * CAMPRE * sizeof(T); }

  // ------------------------------------------------------------------ kernel launchers
  // The LM kernels read the current / trial buffers through a device-resident pointer table that k_decide swaps on
  // acceptance; the host mirror `cur` is refreshed from LMState::cur whenever the state is read back.
  ParamSets<T> psets{};              // both buffer sets + the side that was current at lm_begin
  void push_ptrs() {                 // (re)build the by-value kernel argument for the host's notion of `cur`
    for (int b = 0; b < 2; ++b) { psets.cams[b] = cams[b].p; psets.pts[b] = pts[b].p; psets.ptsT[b] = ptsT[b].p; psets.campre[b] = campre[b].p; }
    psets.base = cur;
    psets.loss_delta = (T)loss_delta;
    psets.loss_kind = loss_kind;
    psets.fixed = has_fixed ? pt_fixed_mask.p : nullptr;
  }
  // argument for launches inside the LM loop (side = base ^ LMState::cur) ...
  ParamSets<T> ps_lm() const { return psets; }
  // ... and for launches outside it (st == nullptr): the host's current side
  ParamSets<T> ps_now() const { ParamSets<T> q = psets; q.base = cur; return q; }
  void launch_residual(T2* r_out) {
    const int g = (int)((M + PM_BLOCK - 1) / PM_BLOCK);
    if (g == 0) return;
    hipLaunchKernelGGL(k_residual<T>, dim3(g), dim3(PM_BLOCK), lds_cams(), stream, campre[cur].p, C, ptsT[cur].p,
                       uv_pm.p, has_w ? w_pm.p : nullptr, ci_pm.p, pi_pm.p, M, r_out, cost_part.p, RLoss<T>{(T)loss_delta, loss_kind});
  }
  void launch_resjac(T2* r_out) {
    const int g = (int)((M + PM_BLOCK - 1) / PM_BLOCK);
    if (g == 0) return;
    const size_t lds = (((size_t)C * CAMPRE + 3) & ~(size_t)3) * sizeof(T) + (size_t)PM_BLOCK * (2 * NCP + 7) * sizeof(T);
    hipLaunchKernelGGL(k_resjac<T>, dim3(g), dim3(PM_BLOCK), lds, stream, campre[cur].p, C, ptsT[cur].p, uv_pm.p,
                       has_w ? w_pm.p : nullptr, ci_pm.p, pi_pm.p, M, r_out, Jc_pm.p, Jp_pm.p);
  }
  // st == nullptr: unconditional (used outside the LM loop); otherwise the launch is a no-op once the solve has
  // terminated or when the last step was rejected and nothing has to be re-linearized
  bool linp_folds_pf(const LMState* st) const { return st != nullptr && linp_lw != 0 && linp_pf && h_state && h_state->free_cams && !lin_pts() && !fused(); }
  void launch_linearize_points(const LMState* st) {
    if (nblk == 0) return;
    if (linp_lw) {
      const uint16_t* tm = dense ? (const uint16_t*)nullptr : grp_mask.p;
      const int32_t* ts = dense ? (const int32_t*)nullptr : grp_start.p;
      // inside the LM loop with free cameras the launch also forms the trial's point factors (k_point_factor's work: launch_schur skips it)
      T* pfp = linp_folds_pf(st) ? pfac.p : nullptr;
      const unsigned char* fx = has_fixed ? pt_fixed_mask.p : (const unsigned char*)nullptr;
      if (linp_lw == 64)
        hipLaunchKernelGGL((k_linearize_points_wave<T, 64>), dim3(nlinp), dim3(PM_BLOCK), 0, stream, st ? ps_lm() : ps_now(), st, C, uv_pm.p,
                           has_w ? w_pm.p : nullptr, N, tm, ts, V.p, gp.p, D2p.p, cost_part.p, gmax_part.p, pfp, fx);
      else
        hipLaunchKernelGGL((k_linearize_points_wave<T, 32>), dim3(nlinp), dim3(PM_BLOCK), 0, stream, st ? ps_lm() : ps_now(), st, C, uv_pm.p,
                           has_w ? w_pm.p : nullptr, N, tm, ts, V.p, gp.p, D2p.p, cost_part.p, gmax_part.p, pfp, fx);
      return;
    }
    const size_t lds = (size_t)PM_BLOCK * 9 * sizeof(double) + lds_cams();
    hipLaunchKernelGGL(k_linearize_points<T>, dim3(nblk), dim3(PM_BLOCK), lds, stream, st ? ps_lm() : ps_now(), st, C,
                       uv_pm.p, has_w ? w_pm.p : nullptr, ci_pm.p, pi_pm.p, pt_start.p, blk_desc.p, V.p, gp.p, D2p.p,
                       cost_part.p, gmax_part.p);
  }
  void launch_linearize_cams(const LMState* st) {
    if (nchunk == 0) return;
    hipLaunchKernelGGL(k_linearize_cams<T>, dim3(nchunk), dim3(256), 0, stream, st ? ps_lm() : ps_now(), st, uv_cm.p,
                       has_w ? w_cm.p : nullptr, pi_cm.p, chunk_cam.p, chunk_begin.p, chunk_end.p, Upart.p);
    hipLaunchKernelGGL(k_reduce_cams, dim3(C), dim3(1024), 0, stream, Upart.p, cam_chunk_start.p, U.p, gc.p, st);
  }
  // the linearisation is folded into the Schur kernel (k_schur_fused) whenever the cameras are free
  bool fused() const { return fused_ok && h_state && h_state->free_cams; }
  bool lin_pts() const { return lin_pts_ok && h_state && h_state->free_cams; }   // f64: points linearised inside k_schur_sym
  int n_lin_parts() const { return (fused() || lin_pts()) ? ksplit : n_linp_blocks(); }       // entries of cost_part / gmax_part
  void launch_schur() {
    if constexpr (sizeof(T) == 4) {
      if (fused() && fused_wide) { launch_schur_wide(); return; }
    }
#if SBA_NCP == 11
    if constexpr (sizeof(T) == 8) {
      if (fused() && fused_wide) {
        double* gm_out = nullptr;
        const FusedDecide fd = make_fused_decide(gm_out);
        const bool tables = !dense;
        const uint16_t* tmask = tables ? grp_mask.p : nullptr;
        const int32_t* tstart = tables ? grp_start.p : nullptr;
        auto go = [&](auto ntw_c, auto pw_c, auto ts_c) {
          constexpr int NTW = decltype(ntw_c)::value, PW = decltype(pw_c)::value, TS = decltype(ts_c)::value;
          constexpr size_t lds64 = SchurWide64Cfg<NTW, PW, TS>::LDS_BYTES;
          hipLaunchKernelGGL((k_schur_fused_wide_f64<NTW, PW, TS>), dim3(ksplit, TS), dim3(SCHUR_THREADS), lds64, stream,
                             ps_lm(), fd, C, uv_pm.p, has_w ? w_pm.p : nullptr, tmask, tstart, N, ksplit, D2p.p, gp.p, pfac.p, slabs.p, bpart.p,
                             gdpart.p, cost_part.p, gm_out, (schur_debug && schur_debug_skip == 0) ? schur_dbg.p : nullptr);
        };
        using P2 = std::integral_constant<int, 2>;
        using P3 = std::integral_constant<int, 3>;
        using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>;
        switch (wide_ntw(C)) {
          case 12: if (wide_pw == 3) go(std::integral_constant<int, 12>{}, P3{}, S1{}); else go(std::integral_constant<int, 12>{}, P2{}, S1{}); break;
          case 13: if (wide_pw == 3) go(std::integral_constant<int, 13>{}, P3{}, S1{}); else go(std::integral_constant<int, 13>{}, P2{}, S1{}); break;
          case 14: go(std::integral_constant<int, 14>{}, P2{}, S2{}); break;
          case 15: go(std::integral_constant<int, 15>{}, P2{}, S2{}); break;
          default: go(std::integral_constant<int, 16>{}, P2{}, S2{}); break;
        }
        d_state.p = fd.st_out;
        pending_decide = false;
        gmax_cur = gm_out;
        if (schur_debug && schur_debug_skip > 0) { --schur_debug_skip; return; }
        if (schur_debug) {
          std::vector<long long> st(64);
          HIPCHK(hipMemcpyAsync(st.data(), schur_dbg.p, 64 * sizeof(long long), hipMemcpyDeviceToHost, stream));
          sync();
          fprintf(stderr, "[schur_fused_wide_f64 stamps, cycles since the first producer stamp; per chunk of 8 or 12 points: producer-done consumer-done]\n");
          for (int i = 0; i < 14; ++i) fprintf(stderr, "  it %2d: P %7lld  C %7lld\n", i, st[2 * i] - st[0], st[2 * i + 1] - st[0]);
          fprintf(stderr, "  phases (cycles): prologue %lld | main loop %lld | fold U %lld | slab stores %lld | tail %lld | whole kernel %lld\n",
                  st[49] - st[48], st[50] - st[49], st[51] - st[50], st[52] - st[51], st[53] - st[52], st[53] - st[48]);
          schur_debug = false;
        }
        return;
      }
      if (fused() && fused_f64) {
        double* gm_out = nullptr;
        const FusedDecide fd = make_fused_decide(gm_out);
        hipLaunchKernelGGL(k_schur_fused_f64, dim3(ksplit), dim3(SCHUR_THREADS), SchurF64Cfg::LDS_BYTES, stream,
                           ps_lm(), fd, C, uv_pm.p, has_w ? w_pm.p : nullptr, pt_start.p, fused_masked ? vis_mask.p : (const uint16_t*)nullptr,
                           N, ksplit, D2p.p, gp.p, pfac.p, slabs.p, bpart.p, gdpart.p, cost_part.p, gm_out,
                           (schur_debug && schur_debug_skip == 0) ? schur_dbg.p : nullptr);
        d_state.p = fd.st_out;
        pending_decide = false;
        gmax_cur = gm_out;
        if (schur_debug && schur_debug_skip > 0) { --schur_debug_skip; return; }
        if (schur_debug) {
          std::vector<long long> st(64);
          HIPCHK(hipMemcpyAsync(st.data(), schur_dbg.p, 64 * sizeof(long long), hipMemcpyDeviceToHost, stream));
          sync();
          fprintf(stderr, "[schur_fused_f64 stamps, cycles since the first producer stamp; per chunk: producer-done consumer-done]\n");
          for (int i = 0; i < 14; ++i) fprintf(stderr, "  it %2d: P %7lld  C %7lld\n", i, st[2 * i] - st[0], st[2 * i + 1] - st[0]);
          fprintf(stderr, "  phases (cycles): prologue %lld | main loop %lld | fold U %lld | slab stores %lld | tail %lld | whole kernel %lld\n",
                  st[49] - st[48], st[50] - st[49], st[51] - st[50], st[52] - st[51], st[53] - st[52], st[53] - st[48]);
          schur_debug = false;
        }
        return;
      }
    }
#endif
#if SBA_NCP == 11
    if constexpr (sizeof(T) == 4) {
      if (fused()) {
        if (fused_bf3) {
          FusedDecide fd{};
          fd.st_in = fd.st_out = d_state.p;
          double* gm_out = (gmax_cur == gmax_part.p) ? gmax_alt.p : gmax_part.p;      // never the array a decision may still read
          if (pending_decide) {
            st_slot ^= 1;
            fd.st_out = d_state_buf.p + st_slot;
            fd.do_decide = 1;
            fd.scal_all = pend_scal; fd.n_ranks = pend_ranks;
            fd.trial_part = trial_part.p; fd.gmax_in = gmax_cur;
            fd.n_trial = n_trial_parts(); fd.n_gmax = n_lin_parts();
            fd.log = reinterpret_cast<LMLogRow*>(d_log.p); fd.log_cap = LOG_CAP;
          }
          hipLaunchKernelGGL(k_schur_fused_bf3, dim3(ksplit), dim3(SCHUR_THREADS), SchurBf3Cfg::LDS_BYTES, stream,
                             ps_lm(), fd, C, uv_pm.p, has_w ? w_pm.p : nullptr, pt_start.p, fused_masked ? vis_mask.p : (const uint16_t*)nullptr,
                             N, ksplit, D2p.p, gp.p, pfac.p,
                             slabs.p, bpart.p, gdpart.p, cost_part.p, gm_out, (schur_debug && schur_debug_skip == 0) ? schur_dbg.p : nullptr,
                             schur_exp);
          d_state.p = fd.st_out;
          pending_decide = false;
          gmax_cur = gm_out;
        } else
          hipLaunchKernelGGL(k_schur_fused, dim3(ksplit), dim3(SCHUR_THREADS), SchurFusedCfg<float>::LDS_BYTES, stream,
                             ps_lm(), d_state.p, C, uv_pm.p, has_w ? w_pm.p : nullptr, pt_start.p, fused_masked ? vis_mask.p : (const uint16_t*)nullptr,
                             N, ksplit, D2p.p, gp.p, pfac.p,
                             slabs.p, bpart.p, gdpart.p, cost_part.p, gmax_part.p, schur_debug ? schur_dbg.p : nullptr);
        if (schur_debug && schur_debug_skip > 0) { --schur_debug_skip; return; }
        if (schur_debug) {
          std::vector<long long> st(64);
          HIPCHK(hipMemcpyAsync(st.data(), schur_dbg.p, 64 * sizeof(long long), hipMemcpyDeviceToHost, stream));
          sync();
          fprintf(stderr, "[schur_fused stamps, cycles since the first producer stamp; per chunk: producer-done consumer-done]\n");
          for (int i = 0; i < 14; ++i) fprintf(stderr, "  it %2d: P %7lld  C %7lld\n", i, st[2 * i] - st[0], st[2 * i + 1] - st[0]);
          fprintf(stderr, "  phases (cycles): prologue %lld | main loop %lld | fold U %lld | slab stores %lld | tail %lld | whole kernel %lld\n",
                  st[49] - st[48], st[50] - st[49], st[51] - st[50], st[52] - st[51], st[53] - st[52], st[53] - st[48]);
          if (fused_bf3)
            fprintf(stderr, "  prologue (cycles): loads requested + LDS zeroed %lld | first barrier passed (record arrived) %lld | decision taken %lld | camera table in LDS %lld | producers set up %lld\n",
                    st[54] - st[48], st[55] - st[48], st[57] - st[48], st[56] - st[48], st[49] - st[48]);
          if (fused_bf3 && st[58])
            fprintf(stderr, "  decision (cycles since kernel start): partial sums folded %lld | record updated %lld\n", st[58] - st[48], st[59] - st[48]);
          schur_debug = false;
        }
        return;
      }
    }
#endif
    if constexpr (SCHUR_LIN_OK<T>) {
      if (lin_pts()) {
        using CfgD = SchurSel<T, true>;
        hipLaunchKernelGGL((k_schur_sym<T, true, false, true>), dim3(ksplit, 1, CfgD::TS), dim3(CfgD::THREADS), CfgD::LDS_BYTES,
                           stream, ps_lm(), d_state.p, C, uv_pm.p, has_w ? w_pm.p : nullptr, ci_pm.p, pi_pm.p, pt_start.p, N,
                           pfac.p, pair_ga.p, pair_gb.p, 0, ksplit, 1, slabs.p, bpart.p, (long long*)nullptr,
                           dense_one_group ? (const uint16_t*)nullptr : vis_mask.p, D2p.p, gp.p, cost_part.p, gmax_part.p);
        return;
      }
    }
    if (N > 0 && !linp_folds_pf(d_state.p))
      hipLaunchKernelGGL(k_point_factor<T>, dim3((N + 255) / 256), dim3(256), 0, stream, V.p, gp.p, D2p.p, d_state.p, N, pfac.p,
                         has_fixed ? pt_fixed_mask.p : (const unsigned char*)nullptr);
    // pairs are stored diagonal ones first: [0, ngroups) are (g,g); the rest are (ga<gb)
    // a camera count that is not a multiple of 16 leaves the last group's panel mostly empty: the PARTIAL
    // instantiations skip the MFMAs of empty tiles (kept apart so that the full-group kernels pay nothing for it)
    // (f64 only: the f32 consumers are paced by their producers and lose more to the per-tile branches than they save)
    if constexpr (SCHUR_SYM<T>) {
      if (C % GROUP_CAMS != 0) { launch_schur_kernels<true>(); return; }
    }
    launch_schur_kernels<false>();
  }
  FusedDecide make_fused_decide(double*& gm_out) {
    FusedDecide fd{};
    fd.st_in = fd.st_out = d_state.p;
    gm_out = (gmax_cur == gmax_part.p) ? gmax_alt.p : gmax_part.p;      // never the array a decision may still read
    if (pending_decide) {
      st_slot ^= 1;
      fd.st_out = d_state_buf.p + st_slot;
      fd.do_decide = 1;
      fd.scal_all = pend_scal; fd.n_ranks = pend_ranks;
      fd.trial_part = trial_part.p; fd.gmax_in = gmax_cur;
      fd.n_trial = n_trial_parts(); fd.n_gmax = n_lin_parts();
      fd.log = reinterpret_cast<LMLogRow*>(d_log.p); fd.log_cap = LOG_CAP;
    }
    return fd;
  }
  void launch_schur_wide() {
    if constexpr (sizeof(T) == 4) {
      double* gm_out = nullptr;
      const FusedDecide fd = make_fused_decide(gm_out);
      // sparse rigs: per (16-camera group, point) visibility mask + index of the point's first observation in the group -- the
      // k_group_index tables with several groups, the one-group mask and the point's start otherwise
      const bool tables = !dense;
      const uint16_t* tmask = !tables ? nullptr : C > GROUP_CAMS ? grp_mask.p : vis_mask.p;
      const int32_t* tstart = !tables ? nullptr : C > GROUP_CAMS ? grp_start.p : pt_start.p;
      auto go = [&](auto ntw_c, auto pw_c) {
        constexpr int NTW = decltype(ntw_c)::value, PW = decltype(pw_c)::value;
        constexpr size_t lds = SchurWideCfg<NTW, PW>::LDS_BYTES;
        hipLaunchKernelGGL((k_schur_fused_wide<NTW, PW>), dim3(ksplit), dim3(SCHUR_THREADS), lds, stream,
                           ps_lm(), fd, C, uv_pm.p, has_w ? w_pm.p : nullptr, tmask, tstart, N, ksplit, D2p.p, gp.p, pfac.p, slabs.p, bpart.p, gdpart.p,
                           cost_part.p, gm_out, (schur_debug && schur_debug_skip == 0) ? schur_dbg.p : nullptr);
      };
      using P2 = std::integral_constant<int, 2>;
      using P3 = std::integral_constant<int, 3>;
      switch (wide_ntw(C)) {
        case 8: if (wide_pw == 3) go(std::integral_constant<int, 8>{}, P3{}); else go(std::integral_constant<int, 8>{}, P2{}); break;
        case 12: if (wide_pw == 3) go(std::integral_constant<int, 12>{}, P3{}); else go(std::integral_constant<int, 12>{}, P2{}); break;
        case 13: if (wide_pw == 3) go(std::integral_constant<int, 13>{}, P3{}); else go(std::integral_constant<int, 13>{}, P2{}); break;
        case 14: go(std::integral_constant<int, 14>{}, P2{}); break;
        case 15: go(std::integral_constant<int, 15>{}, P2{}); break;
        default: go(std::integral_constant<int, 16>{}, P2{}); break;
      }
      d_state.p = fd.st_out;
      pending_decide = false;
      gmax_cur = gm_out;
      if (schur_debug && schur_debug_skip > 0) { --schur_debug_skip; return; }
      if (schur_debug) {
        std::vector<long long> st(64);
        HIPCHK(hipMemcpyAsync(st.data(), schur_dbg.p, 64 * sizeof(long long), hipMemcpyDeviceToHost, stream));
        sync();
        fprintf(stderr, "[schur_fused_wide stamps, cycles since the first producer stamp; per round: producer-done consumer-done]\n");
        for (int i = 0; i < 14; ++i) fprintf(stderr, "  it %2d: P %7lld  C %7lld\n", i, st[2 * i] - st[0], st[2 * i + 1] - st[0]);
        fprintf(stderr, "  phases (cycles): prologue %lld | main loop %lld | fold U %lld | slab stores %lld | tail %lld | whole kernel %lld\n",
                st[49] - st[48], st[50] - st[49], st[51] - st[50], st[52] - st[51], st[53] - st[52], st[53] - st[48]);
        schur_debug = false;
      }
    }
  }
  template <bool PARTIAL> void launch_schur_kernels() {
    using CfgD = SchurSel<T, true>;
    using CfgO = SchurSel<T, false>;
    if constexpr (SCHUR_SYM<T>) {
      hipLaunchKernelGGL((k_schur_sym<T, true, PARTIAL>), dim3(ksplit, ngroups, CfgD::TS), dim3(CfgD::THREADS), CfgD::LDS_BYTES,
                         stream, ps_lm(), d_state.p, C, uv_pm.p, has_w ? w_pm.p : nullptr, ci_pm.p, pi_pm.p, pt_start.p, N,
                         pfac.p, pair_ga.p, pair_gb.p, 0, ksplit, (int)dense, slabs.p, bpart.p, schur_debug ? schur_dbg.p : nullptr,
                         (const uint16_t*)nullptr, (double*)nullptr, (double*)nullptr, (double*)nullptr, (double*)nullptr,
                         grp_indexed ? grp_mask.p : (const uint16_t*)nullptr, grp_indexed ? grp_start.p : (const int32_t*)nullptr);
      if (schur_debug) {
        std::vector<long long> st(64);
        HIPCHK(hipMemcpyAsync(st.data(), schur_dbg.p, 64 * sizeof(long long), hipMemcpyDeviceToHost, stream));
        sync();
        fprintf(stderr, "[schur_sym stamps, cycles; per chunk: produce-done after-barrier consume-done]\n");
        for (int i = 0; i < 8; ++i)
          fprintf(stderr, "  it %2d: P %7lld  B %7lld  C %7lld\n", i, st[3 * i] - st[0], st[3 * i + 1] - st[0], st[3 * i + 2] - st[0]);
        schur_debug = false;
      }
      if (npairs > ngroups)
        hipLaunchKernelGGL((k_schur_sym<T, false, PARTIAL>), dim3(ksplit, npairs - ngroups, CfgO::TS), dim3(CfgO::THREADS),
                           CfgO::LDS_BYTES, stream, ps_lm(), d_state.p, C, uv_pm.p,
                           has_w ? w_pm.p : nullptr, ci_pm.p, pi_pm.p, pt_start.p, N, pfac.p,
                           pair_ga.p, pair_gb.p, ngroups, ksplit, (int)dense, slabs.p, bpart.p, nullptr,
                           (const uint16_t*)nullptr, (double*)nullptr, (double*)nullptr, (double*)nullptr, (double*)nullptr,
                           grp_indexed ? grp_mask.p : (const uint16_t*)nullptr, grp_indexed ? grp_start.p : (const int32_t*)nullptr);
      return;
    } else {
    if (diag_pairs_bf3()) {
      hipLaunchKernelGGL(k_schur_diag_bf3, dim3(ksplit, ngroups), dim3(SCHUR_THREADS), SchurPairCfg::LDS_BYTES, stream,
                         ps_lm(), d_state.p, C, uv_pm.p, has_w ? w_pm.p : nullptr, grp_mask.p, grp_start.p, N, pfac.p, pair_ga.p, 0, ksplit,
                         slabs.p, bpart.p, pairs_fold_u ? gdpart.p : (double*)nullptr);
    } else
    hipLaunchKernelGGL((k_schur<T, true, PARTIAL>), dim3(ksplit, ngroups, CfgD::TS), dim3(CfgD::THREADS), CfgD::LDS_BYTES,
                       stream, ps_lm(), d_state.p, C, uv_pm.p, has_w ? w_pm.p : nullptr, ci_pm.p, pi_pm.p, pt_start.p, N,
                       pfac.p, pair_ga.p, pair_gb.p, 0, ksplit, (int)dense, slabs.p, bpart.p,
                       schur_debug ? schur_dbg.p : nullptr, grp_indexed ? grp_mask.p : (const uint16_t*)nullptr,
                       grp_indexed ? grp_start.p : (const int32_t*)nullptr);
    if (schur_debug) {
      std::vector<long long> st(64);
      HIPCHK(hipMemcpyAsync(st.data(), schur_dbg.p, 64 * sizeof(long long), hipMemcpyDeviceToHost, stream));
      sync();
      fprintf(stderr, "[schur stamps, cycles since first barrier; per chunk: producer-done consumer-done barrier-out]\n");
      for (int i = 0; i < 14; ++i)
        fprintf(stderr, "  it %2d: P %7lld  C %7lld  out %7lld\n", i, st[3 * i] - st[2], st[3 * i + 1] - st[2], st[3 * i + 2] - st[2]);
      schur_debug = false;
    }
    if (npairs > ngroups && offdiag_pairs_bf3()) {
      hipLaunchKernelGGL(k_schur_offdiag_bf3, dim3(ksplit, npairs - ngroups), dim3(SCHUR_THREADS), SchurBf3OffCfg::LDS_BYTES, stream,
                         ps_lm(), d_state.p, C, uv_pm.p, has_w ? w_pm.p : nullptr, grp_mask.p, grp_start.p, N, pfac.p, pair_ga.p, pair_gb.p,
                         ngroups, ksplit, slabs.p);
    } else
    if (npairs > ngroups)
      hipLaunchKernelGGL((k_schur<T, false, PARTIAL>), dim3(ksplit, npairs - ngroups, CfgO::TS), dim3(CfgO::THREADS),
                         CfgO::LDS_BYTES, stream, ps_lm(), d_state.p, C, uv_pm.p,
                         has_w ? w_pm.p : nullptr, ci_pm.p, pi_pm.p, pt_start.p, N, pfac.p,
                         pair_ga.p, pair_gb.p, ngroups, ksplit, (int)dense, slabs.p, bpart.p, nullptr,
                         grp_indexed ? grp_mask.p : (const uint16_t*)nullptr, grp_indexed ? grp_start.p : (const int32_t*)nullptr);
    }
  }
  // dense visibility with one camera group: the row-reduction kernel (any dtype); its partial rows are per workgroup
  bool backsub_masked = false;       // sparse one-group rig dense enough for the lane = (point, camera) back substitution (visibility mask)
  bool backsub_wide = false;         // 17 .. 23 cameras: k_backsub_dense<T, 32> / <T, 0>
  bool backsub_pack = false;
  bool backsub_wave = false;         // 33 .. 128 cameras: k_backsub_dense<T, 64>
  int linp_lw = 0;                   // 24 .. 128 cameras with (point, camera) tables: k_linearize_points_wave<T, 32 / 64>; 0 = k_linearize_points
  bool linp_pf = true;               // ... which also forms the point factors (SBA_LINP_PF=0: k_point_factor in a launch of its own)
  int nlinp = 0;                     // ... its (persistent) workgroups = entries of cost_part / gmax_part
  int n_linp_blocks() const { return linp_lw ? nlinp : nblk; }
  bool backsub_dense() const { return dense_one_group || backsub_masked || backsub_wide || backsub_wave; }
  int n_trial_parts() const { return backsub_dense() ? nbs_dense : nblk; }
  void launch_backsub_trial() {
    if (nblk == 0) return;
    if (backsub_pack) {
      hipLaunchKernelGGL((k_backsub_dense<T, 0>), dim3(nbs_dense), dim3(PM_BLOCK), 0, stream, ps_lm(), C, uv_pm.p,
                         has_w ? w_pm.p : nullptr, N, pfac.p, gp.p, D2p.p, delta_c.p, d_state.p, trial_part.p, nbs_dense,
                         dense ? (const uint16_t*)nullptr : grp_mask.p, dense ? (const int32_t*)nullptr : grp_start.p);
      return;
    }
    if (backsub_wave) {
      hipLaunchKernelGGL((k_backsub_dense<T, 64>), dim3(nbs_dense), dim3(PM_BLOCK), 0, stream, ps_lm(), C, uv_pm.p,
                         has_w ? w_pm.p : nullptr, N, pfac.p, gp.p, D2p.p, delta_c.p, d_state.p, trial_part.p, nbs_dense,
                         dense ? (const uint16_t*)nullptr : grp_mask.p, dense ? (const int32_t*)nullptr : grp_start.p);
      return;
    }
    if (backsub_wide) {
      hipLaunchKernelGGL((k_backsub_dense<T, 32>), dim3(nbs_dense), dim3(PM_BLOCK), 0, stream, ps_lm(), C, uv_pm.p,
                         has_w ? w_pm.p : nullptr, N, pfac.p, gp.p, D2p.p, delta_c.p, d_state.p, trial_part.p, nbs_dense,
                         dense ? (const uint16_t*)nullptr : grp_mask.p, dense ? (const int32_t*)nullptr : grp_start.p);
      return;
    }
    if (backsub_dense()) {
      hipLaunchKernelGGL(k_backsub_dense<T>, dim3(nbs_dense), dim3(PM_BLOCK), 0, stream, ps_lm(), C, uv_pm.p,
                         has_w ? w_pm.p : nullptr, N, pfac.p, gp.p, D2p.p, delta_c.p, d_state.p, trial_part.p, nbs_dense,
                         dense_one_group ? (const uint16_t*)nullptr : vis_mask.p, dense_one_group ? (const int32_t*)nullptr : pt_start.p);
      return;
    }
    const size_t lds = (size_t)PM_BLOCK * 6 * sizeof(double) + (2 * (size_t)C * CAMPRE + (size_t)C * NCP) * sizeof(T);
    hipLaunchKernelGGL(k_backsub_trial<T>, dim3(nblk), dim3(PM_BLOCK), lds, stream, ps_lm(), C,
                       uv_pm.p, has_w ? w_pm.p : nullptr, ci_pm.p, pi_pm.p, pt_start.p, blk_desc.p,
                       pfac.p, gp.p, D2p.p, delta_c.p, d_state.p, trial_part.p, nblk);
  }

  // ------------------------------------------------------------------ model evaluation entry points
  int residual(const double* x, double* r_out, double* cost_out) override {
    if (!uploaded) { err = "sba_upload has not been called"; return SBA_ERR_STATE; }
    HIPCHK(hipSetDevice(device));
    if (x) set_params(x, x + (size_t)C * NCP);
    if (r_out && r_pm.n != (size_t)M) r_pm.alloc(M);
    launch_residual(r_out ? r_pm.p : nullptr);
    const int g = (int)((M + PM_BLOCK - 1) / PM_BLOCK);
    std::vector<double> part_v((size_t)g > LAND_DOUBLES ? g : 0);
    double* part = part_v.empty() ? h_land : part_v.data();
    if (g) HIPCHK(hipMemcpyAsync(part, cost_part.p, sizeof(double) * g, hipMemcpyDeviceToHost, stream));
    std::vector<T2> r(r_out ? M : 0);
    if (r_out && M) HIPCHK(hipMemcpyAsync(r.data(), r_pm.p, sizeof(T2) * M, hipMemcpyDeviceToHost, stream));
    sync();
    HIPCHK(hipGetLastError());
    double c = 0;
    for (int i = 0; i < g; ++i) c += part[i];
    if (cost_out) *cost_out = c;
    if (r_out)
      par_for(M, [&](int64_t lo, int64_t hi, int) {
        for (int64_t k = lo; k < hi; ++k) { const int64_t i = identity_perm ? k : perm[k]; r_out[2 * i] = (double)r[k].x; r_out[2 * i + 1] = (double)r[k].y; }
      });
    return SBA_OK;
  }

  int residual_jacobian(const double* x, double* r_out, double* Jc_out, double* Jp_out) override {
    if (!uploaded) { err = "sba_upload has not been called"; return SBA_ERR_STATE; }
    HIPCHK(hipSetDevice(device));
    if (x) set_params(x, x + (size_t)C * NCP);
    if (r_pm.n != (size_t)M) r_pm.alloc(M);
    if (Jc_pm.n != (size_t)M * 2 * NCP) { Jc_pm.alloc((size_t)M * 2 * NCP); Jp_pm.alloc((size_t)M * 6); }
    launch_resjac(r_pm.p);
    std::vector<T2> r(M);
    std::vector<T> jc((size_t)M * 2 * NCP), jp((size_t)M * 6);
    if (M) {
      HIPCHK(hipMemcpyAsync(r.data(), r_pm.p, sizeof(T2) * M, hipMemcpyDeviceToHost, stream));
      HIPCHK(hipMemcpyAsync(jc.data(), Jc_pm.p, sizeof(T) * jc.size(), hipMemcpyDeviceToHost, stream));
      HIPCHK(hipMemcpyAsync(jp.data(), Jp_pm.p, sizeof(T) * jp.size(), hipMemcpyDeviceToHost, stream));
    }
    sync();
    HIPCHK(hipGetLastError());
    for (int64_t k = 0; k < M; ++k) {
      const int64_t i = identity_perm ? k : perm[k];
      if (r_out) { r_out[2 * i] = (double)r[k].x; r_out[2 * i + 1] = (double)r[k].y; }
      if (Jc_out) for (int e = 0; e < 2 * NCP; ++e) Jc_out[(size_t)i * 2 * NCP + e] = (double)jc[(size_t)k * 2 * NCP + e];
      if (Jp_out) for (int e = 0; e < 6; ++e) Jp_out[(size_t)i * 6 + e] = (double)jp[(size_t)k * 6 + e];
    }
    return SBA_OK;
  }

  // ------------------------------------------------------------------ squared-pixel-error variants (pySBA.py:151-205)
  bool sq_mode() const { return opts.mode == SBA_MODE_CAMS_ONLY_SQ || opts.mode == SBA_MODE_TRANSFORM_SQ; }
  int nblk_sq = 0, nchunk_sq = 0;
  DevBuf<double> sq_part, sq_16, theta[2];
  DevBuf<int32_t> sq_start;
  double h_theta[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  ThetaSets tsets() const { ThetaSets t; t.th[0] = theta[0].p; t.th[1] = theta[1].p; t.base = cur_at_begin; return t; }
  void sq_setup() {
    nblk_sq = (int)((M + PM_BLOCK - 1) / PM_BLOCK);
    std::vector<int32_t> st;
    if (opts.mode == SBA_MODE_CAMS_ONLY_SQ) {          // one 16x16 per camera, from the camera-major chunks
      nchunk_sq = nchunk;
      sq_16.alloc((size_t)C * 256);
    } else {                                           // one 16x16 for the whole problem
      nchunk_sq = (int)((M + SQ_CHUNK - 1) / SQ_CHUNK);
      st = {0, nchunk_sq};
      sq_start.upload(st, stream);
      sq_16.alloc(256);
      for (int b = 0; b < 2; ++b) if (theta[b].n != 12) theta[b].alloc(12);
    }
    sq_part.alloc((size_t)std::max(nchunk_sq, 1) * 256);
    if (trial_part.n < (size_t)4 * std::max(nblk_sq, 1)) trial_part.alloc((size_t)4 * std::max(nblk_sq, 1));
    if (gmax_part.n < (size_t)std::max(nblk_sq, 1)) gmax_part.alloc(std::max(nblk_sq, 1));
    gmax_part.zero(stream);
  }
  void launch_sq_linearize(const LMState* st) {
    if (nchunk_sq == 0) return;
    const ParamSets<T> ps = st ? ps_lm() : ps_now();
    ThetaSets ts = tsets();
    if (!st) ts.base = cur;
    if (opts.mode == SBA_MODE_CAMS_ONLY_SQ) {
      const size_t lds = ((size_t)4 * 16 * 130 + CAMPRE) * sizeof(T);
      hipLaunchKernelGGL((k_sq_linearize<T, 1>), dim3(nchunk_sq), dim3(256), lds, stream, ps, ts, st, C, uv_cm.p,
                         has_w ? w_cm.p : nullptr, (const int32_t*)nullptr, pi_cm.p, chunk_cam.p, chunk_begin.p, chunk_end.p, M, sq_part.p);
      hipLaunchKernelGGL(k_reduce16, dim3(C), dim3(1024), 0, stream, sq_part.p, cam_chunk_start.p, sq_16.p, st);
    } else {
      const size_t lds = ((size_t)4 * 16 * 130 + (size_t)C * CAMPRE) * sizeof(T);
      hipLaunchKernelGGL((k_sq_linearize<T, 2>), dim3(nchunk_sq), dim3(256), lds, stream, ps, ts, st, C, uv_pm.p,
                         has_w ? w_pm.p : nullptr, ci_pm.p, pi_pm.p, (const int32_t*)nullptr, (const int32_t*)nullptr,
                         (const int32_t*)nullptr, M, sq_part.p);
      hipLaunchKernelGGL(k_reduce16, dim3(1), dim3(1024), 0, stream, sq_part.p, sq_start.p, sq_16.p, st);
    }
  }
  void launch_sq_trial() {
    if (nblk_sq == 0) return;
    if (opts.mode == SBA_MODE_CAMS_ONLY_SQ)
      hipLaunchKernelGGL((k_sq_trial<T, 1>), dim3(nblk_sq), dim3(PM_BLOCK), lds_cams(), stream, ps_lm(), tsets(), d_state.p, C,
                         uv_pm.p, has_w ? w_pm.p : nullptr, ci_pm.p, pi_pm.p, M, trial_part.p, nblk_sq);
    else
      hipLaunchKernelGGL((k_sq_trial<T, 2>), dim3(nblk_sq), dim3(PM_BLOCK), lds_cams(), stream, ps_lm(), tsets(), d_state.p, C,
                         uv_pm.p, has_w ? w_pm.p : nullptr, ci_pm.p, pi_pm.p, M, trial_part.p, nblk_sq);
  }
  // cost (0.5 sum rho^2) and max |gradient| from the 16x16 sums of the last squared-variant linearization
  void sq_read(double& cost, double& gmax) {
    const int ng = (opts.mode == SBA_MODE_CAMS_ONLY_SQ) ? C : 1;
    const int np = (opts.mode == SBA_MODE_CAMS_ONLY_SQ) ? NCP : 12;
    std::vector<double> h((size_t)ng * 256);
    HIPCHK(hipMemcpyAsync(h.data(), sq_16.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, stream));
    sync();
    cost = 0; gmax = 0;
    for (int g = 0; g < ng; ++g) {
      cost += 0.5 * h[(size_t)g * 256 + np * 16 + np];
      for (int i = 0; i < np; ++i) gmax = std::max(gmax, std::fabs(h[(size_t)g * 256 + i * 16 + np]));
    }
  }

  // shared intrinsics (pySBA.py:252-325): unknowns [f,k1,k2 | 6 extrinsics x C | 2 centre x C], pySBA.py:313 order
  // (tangential model: the per-camera tail is [p1, p2, cx, cy])
  void build_tie_tables() {
    constexpr int PC = NCP - 9;          // per-camera intrinsics that stay free: cx, cy (+ p1, p2 in the tangential model)
    n_tied = 3 + (6 + PC) * C;
    std::vector<int32_t> tie(n), start(n_tied + 1, 0), idx(n), firstv(n_tied);
    for (int c = 0; c < C; ++c)
      for (int e = 0; e < NCP; ++e)
        tie[c * NCP + e] = e < 6 ? 3 + 6 * c + e : e < 9 ? e - 6 : 3 + 6 * C + PC * c + (e - 9);
    for (int i = 0; i < n; ++i) start[tie[i] + 1]++;
    for (int a = 0; a < n_tied; ++a) start[a + 1] += start[a];
    std::vector<int32_t> fill(start.begin(), start.end() - 1);
    for (int i = 0; i < n; ++i) idx[fill[tie[i]]++] = i;
    for (int a = 0; a < n_tied; ++a) firstv[a] = idx[start[a]];
    h_tie = tie;
    tie_map.upload(tie, stream); tie_pre_start.upload(start, stream); tie_pre_idx.upload(idx, stream); tie_first.upload(firstv, stream);
    E_tied.alloc((size_t)n_tied * n_tied + 3 * (size_t)n_tied + 1);
    sync();
  }
  int n_tied = 0;
  std::vector<int32_t> h_tie;
  DevBuf<int32_t> tie_map, tie_pre_start, tie_pre_idx, tie_first;
  DevBuf<double> E_tied;

  // ------------------------------------------------------------------ LM phases
  int64_t exchange_size() const override { return (int64_t)n * n + 3 * (int64_t)n + 1; }
  static constexpr int LOG_CAP = 4096;
  static constexpr int BATCH = 4;      // LM iterations enqueued between two host polls of the state

  // ------------------------------------------------------------------ multi-rank plumbing (RCCL on the engine's stream)
  int comm_init(const uint8_t* id, int rank, int n_ranks) override {
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks || !id) { err = "bad communicator arguments"; return SBA_ERR_INVALID; }
    if (ipc_on || ipc_mine) { err = "the handle already has a peer-mapped exchange area (sba_ipc_export): the two exchanges are exclusive"; return SBA_ERR_STATE; }
    Rccl& r = Rccl::get();
    if (!r.ok) { err = r.why; return SBA_ERR_UNSUPPORTED; }
    HIPCHK(hipSetDevice(device));
    if (comm) { (void)r.comm_destroy(comm); comm = nullptr; }
    Rccl::UniqueId uid;
    memcpy(uid.internal, id, Rccl::ID_BYTES);
    RCCLCHK(r.comm_init_rank(&comm, n_ranks, uid, rank));
    comm_rank = rank; comm_n = n_ranks;
    // (a handle may be given a new communicator: the bump arena never returns memory, so only grow)
    if (sc_loc.n < (size_t)NSCAL) sc_loc.alloc(NSCAL);
    if (sc_all.n < (size_t)NSCAL * n_ranks) sc_all.alloc((size_t)NSCAL * n_ranks);
    if (comm_tmp.n < (size_t)std::max(n, 1) + 8) comm_tmp.alloc(std::max(n, 1) + 8);
    if (!h_comm) HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h_comm), sizeof(double) * (std::max(n, 1) + 8), hipHostMallocDefault));
    return SBA_OK;
  }
  // in-place all-reduce of a few host doubles (begin / finish only: the LM loop itself never leaves the device)
  // ------------------------------------------------------------------ one-shot exchange (sba_ipc.hpp)
  void ipc_close_peers() {
    for (size_t r = 0; r < ipc_area.size(); ++r)
      if (ipc_area[r] && r < ipc_opened.size() && ipc_opened[r]) (void)hipIpcCloseMemHandle(ipc_area[r]);
    ipc_area.clear(); ipc_opened.clear();
  }
  int ipc_export(int n_ranks, uint8_t* handle_out) override {
    if (n_ranks < 1 || !handle_out) { err = "bad arguments"; return SBA_ERR_INVALID; }
    if (ipc_on || ipc_mine) { err = "the handle already has an exchange area"; return SBA_ERR_STATE; }
    if (comm) { err = "the handle already has an RCCL communicator (the two exchanges are exclusive)"; return SBA_ERR_STATE; }
    HIPCHK(hipSetDevice(device));
    ipc_L = IpcLayout::make(n);
    if (const char* e = getenv("SBA_IPC_TIMEOUT_S")) {
      char* end = nullptr;
      const double sec = strtod(e, &end);
      if (end != e && sec > 0 && sec < 3600) ipc_timeout_ticks = (long long)(sec * 1e8);
    }
    void* pmem = nullptr;
    // uncached: the peers read what this rank's kernels wrote without a kernel boundary of THEIR stream in between.  A cached
    // allocation would still work between ranks that share one device (one L2) but not between GPUs, so it is never taken
    // silently: SBA_IPC_ALLOW_CACHED=1 asks for it (one-card rehearsals on a stack without the uncached flag) and says so.
    if (hipExtMallocWithFlags(&pmem, ipc_L.total * sizeof(double), hipDeviceMallocUncached) != hipSuccess) {
      (void)hipGetLastError();
      if (!getenv("SBA_IPC_ALLOW_CACHED")) {
        err = "sba_ipc_export: uncached device memory (hipDeviceMallocUncached) is not available; the one-shot exchange needs it between GPUs "
              "(SBA_IPC_ALLOW_CACHED=1 accepts cached memory for ranks that share ONE device)";
        return SBA_ERR_HIP;
      }
      fprintf(stderr, "[sba_ipc] warning: exchange area in CACHED device memory (SBA_IPC_ALLOW_CACHED=1): valid between ranks on one device only\n");
      HIPCHK(hipMalloc(&pmem, ipc_L.total * sizeof(double)));
    }
    ipc_mine = static_cast<double*>(pmem);
    HIPCHK(hipMemsetAsync(ipc_mine, 0, ipc_L.total * sizeof(double), stream));
    sync();
    hipIpcMemHandle_t hm;
    HIPCHK(hipIpcGetMemHandle(&hm, ipc_mine));
    static_assert(sizeof(hm) == SBA_IPC_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
    memcpy(handle_out, &hm, sizeof hm);
    memcpy(ipc_my_handle, &hm, sizeof hm);
    // ranks that live in the SAME process (several handles, one per thread or per device) cannot open each other's handles
    // (hipIpcOpenMemHandle refuses the exporting process): they find the pointer here instead
    IpcLocalAreas::get().add(ipc_my_handle, ipc_mine);
    ipc_planned = n_ranks;
    return SBA_OK;
  }
  int ipc_attach(int rank, int n_ranks, const uint8_t* handles) override {
    if (!ipc_mine || n_ranks != ipc_planned || rank < 0 || rank >= n_ranks || !handles) { err = "sba_ipc_export first, with the same n_ranks"; return SBA_ERR_STATE; }
    if (comm) { err = "the handle already has an RCCL communicator"; return SBA_ERR_STATE; }
    if (ipc_on) { err = "the handle is already attached to its peers' exchange areas"; return SBA_ERR_STATE; }
    HIPCHK(hipSetDevice(device));
    ipc_area.assign(n_ranks, nullptr);
    ipc_opened.assign(n_ranks, 0);
    for (int r = 0; r < n_ranks; ++r) {
      if (r == rank) { ipc_area[r] = ipc_mine; continue; }
      if (double* local = IpcLocalAreas::get().find(handles + (size_t)r * SBA_IPC_HANDLE_BYTES)) { ipc_area[r] = local; continue; }
      hipIpcMemHandle_t hp;
      memcpy(&hp, handles + (size_t)r * SBA_IPC_HANDLE_BYTES, sizeof hp);
      void* q = nullptr;
      const hipError_t oe = hipIpcOpenMemHandle(&q, hp, hipIpcMemLazyEnablePeerAccess);
      if (oe != hipSuccess) {
        (void)hipGetLastError();
        ipc_close_peers();
        err = std::string("hipIpcOpenMemHandle of rank ") + std::to_string(r) + "'s area failed: " + hipGetErrorString(oe);
        return SBA_ERR_HIP;
      }
      ipc_area[r] = static_cast<double*>(q);
      ipc_opened[r] = 1;
    }
    // ranks that share this card: their launches compete for its CUs, and k_chol_big_dag's progress argument (in-order dispatch of
    // ONE launch) does not cover several such launches holding each other's CUs on different XCDs -- they get the per-column launches
    card_shared = false;
    for (int r = 0; r < n_ranks; ++r) {
      if (r == rank) continue;
      hipPointerAttribute_t at;
      if (hipPointerGetAttributes(&at, ipc_area[r]) == hipSuccess) { if (at.device == device) card_shared = true; }
      else (void)hipGetLastError();
    }
    if (chol_debug) fprintf(stderr, "[sba_ipc_attach] rank %d of %d: %s\n", rank, n_ranks, card_shared ? "a peer's exchange area lives on this device: ranks share the card (per-column launches for the large Cholesky)" : "no peer on this device");
    ipc_ptrs.upload(ipc_area, stream);
    if (ipc_fail.n == 0) ipc_fail.alloc(1);
    ipc_fail.zero(stream);
    comm_rank = rank; comm_n = n_ranks;
    if (sc_all.n < (size_t)NSCAL * n_ranks) sc_all.alloc((size_t)NSCAL * n_ranks);
    if (comm_tmp.n < (size_t)std::max(n, 1) + 8) comm_tmp.alloc(std::max(n, 1) + 8);
    if (!h_comm) HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h_comm), sizeof(double) * (std::max(n, 1) + 8), hipHostMallocDefault));
    sync();
    ipc_on = true;
    return SBA_OK;
  }
  void ipc_publish(int kind, unsigned long long seq, const LMState* st) {
    hipLaunchKernelGGL(k_ipc_publish, dim3(1), dim3(1), 0, stream, ipc_mine, ipc_L.flag_of(kind, (int)(seq & 1)), seq, st);
  }
  void ipc_gate(int kind, unsigned long long seq, LMState* st, size_t copy_off, int ncopy, double* dst) {
    hipLaunchKernelGGL(k_ipc_gate, dim3(1), dim3(64), 0, stream, ipc_ptrs.p, comm_n, ipc_L.flag_of(kind, (int)(seq & 1)), seq, st,
                       st ? (int*)nullptr : ipc_fail.p, copy_off, ncopy, dst, ipc_timeout_ticks);
  }
  void comm_reduce(double* v, int count, int op) {
    for (int i = 0; i < count; ++i) h_comm[i] = v[i];
    if (ipc_on) {
      // after a timeout the ranks' exchange counters no longer agree (some passed the gate, some did not): the handle cannot
      // take part in another exchange and has to be destroyed -- on every rank
      if (ipc_dead) throw HipError{hipErrorNotReady, "an earlier exchange of this handle timed out (sba_ipc): destroy it and start over", __FILE__, __LINE__};
      const unsigned long long s = ++ipc_seq[2];
      const size_t slot = ipc_L.vec_slot((int)(s & 1));
      HIPCHK(hipMemcpyAsync(ipc_mine + slot, h_comm, sizeof(double) * count, hipMemcpyHostToDevice, stream));
      ipc_publish(2, s, nullptr);
      ipc_gate(2, s, nullptr, 0, 0, nullptr);
      hipLaunchKernelGGL(k_ipc_reduce_vec, dim3(1), dim3(256), 0, stream, ipc_ptrs.p, comm_n, slot, count, op == Rccl::kMax ? 1 : 0, comm_tmp.p);
      int failed = 0;
      HIPCHK(hipMemcpyAsync(h_comm, comm_tmp.p, sizeof(double) * count, hipMemcpyDeviceToHost, stream));
      HIPCHK(hipMemcpyAsync(&failed, ipc_fail.p, sizeof(int), hipMemcpyDeviceToHost, stream));
      sync();
      if (failed) { ipc_dead = true; throw HipError{hipErrorNotReady, "a peer rank did not reach the exchange in time (sba_ipc; SBA_IPC_TIMEOUT_S, default 5 s)", __FILE__, __LINE__}; }
      for (int i = 0; i < count; ++i) v[i] = h_comm[i];
      return;
    }
    HIPCHK(hipMemcpyAsync(comm_tmp.p, h_comm, sizeof(double) * count, hipMemcpyHostToDevice, stream));
    RCCLCHK(Rccl::get().all_reduce(comm_tmp.p, comm_tmp.p, count, Rccl::kFloat64, op, comm, stream));
    HIPCHK(hipMemcpyAsync(h_comm, comm_tmp.p, sizeof(double) * count, hipMemcpyDeviceToHost, stream));
    sync();
    for (int i = 0; i < count; ++i) v[i] = h_comm[i];
  }
  void comm_sum(double* v, int count) { comm_reduce(v, count, Rccl::kSum); }
  void comm_max(double* v, int count) { comm_reduce(v, count, Rccl::kMax); }

  int lm_begin(const sba_lm_opts* o) override {
    if (!uploaded) { err = "sba_upload has not been called"; return SBA_ERR_STATE; }
    HIPCHK(hipSetDevice(device));
    opts = *o;
    st_slot = 0;
    d_state.p = d_state_buf.p;
    pending_decide = false;
    prof_on = opts.reserved[0] != 0;
    for (int k = 0; k < KP_N; ++k) { prof_us[k] = 0; prof_cnt[k] = 0; }
    pslot = 0;
    for (auto& u : pev_used) u = false;
    if (opts.mode < SBA_MODE_FULL || opts.mode > SBA_MODE_TRANSFORM_SQ) { err = "unsupported mode"; return SBA_ERR_UNSUPPORTED; }
    if (opts.mode == SBA_MODE_SHARED_INTR) build_tie_tables();
    // initial cost; scipy raises ValueError when it is not finite (least_squares.py:844-845)
    double c0 = 0;
    int rc = SBA_OK;
    if (multi() && sq_mode() && comm_n > 1) { err = "the squared-error variants (camonly, transform_points_3d) run on one GPU only"; return SBA_ERR_UNSUPPORTED; }  // (same on every rank)
    // With a communicator a failure on THIS rank must not keep it out of the collective its peers are about to enter: the
    // failure travels as a flag inside that all-reduce and every rank returns an error together.
    try { rc = residual(nullptr, nullptr, &c0); }
    catch (const HipError& e) {
      if (!multi()) throw;
      err = std::string("HIP error in lm_begin: ") + hipGetErrorString(e.e) + " (" + e.what + ")";
      rc = SBA_ERR_HIP;
    }
    if (rc && !multi()) return rc;
    N_global = N;
    if (multi()) {
      // every rank must see the same initial cost (a non-finite one on ANY rank fails the solve on ALL of them, before the
      // first collective of the loop) and the same default evaluation budget, 100 x the GLOBAL number of parameters
      double v[3] = {rc ? 0.0 : c0, (double)N, rc ? 1.0 : 0.0};
      comm_sum(v, 3);
      if (v[2] > 0) {
        if (!rc) { err = "lm_begin failed on " + std::to_string((int)(v[2] + 0.5)) + " peer rank(s)"; rc = SBA_ERR_STATE; }
        return rc;
      }
      c0 = v[0];
      N_global = (long long)(v[1] + 0.5);
    }
    initial_cost = c0;
    LMState s{};
    s.lam = opts.lambda0 > 0 ? opts.lambda0 : 1e-4;
    // x_scale = 1 (unscaled damping): lambda = tau * max diag(J^T J), set on the device.  tau is tiny because the
    // parameters of these variants differ by orders of magnitude in scale (rotation vs focal length; affine matrix vs
    // translation column), and only a near Gauss-Newton step moves the weakly scaled ones -- scipy gets the same
    // effect from its large initial trust radius
    if (sq_mode()) s.lam = -(opts.lambda0 > 0 ? opts.lambda0 : 1e-9);
    s.nu = 2.0;
    s.cost = c0;
    s.ftol = opts.ftol; s.xtol = opts.xtol; s.gtol = opts.gtol;
    s.lam_min = 1e-12; s.lam_max = 1e12;
    s.nfev = 1; s.njev = 1;
    const long long nparam = opts.mode == SBA_MODE_CAMS_ONLY_SQ ? (long long)n : opts.mode == SBA_MODE_TRANSFORM_SQ ? 12LL :
        (opts.mode == SBA_MODE_FULL ? (long long)n : opts.mode == SBA_MODE_SHARED_INTR ? (long long)n_tied : 0) + 3LL * N_global;
    s.max_nfev = opts.max_nfev > 0 ? opts.max_nfev : 100 * nparam;
    s.status = -1; s.fresh = 1; s.need_lin = 1;
    s.always_relin = opts.always_relinearize ? 1 : 0;
    s.max_iter = opts.max_iter > 0 ? opts.max_iter : 0;
    s.cur = 0;
    s.free_cams = (opts.mode == SBA_MODE_POINTS_ONLY) ? 0 : 1;
    *h_state = s;
    HIPCHK(hipMemcpyAsync(d_state.p, h_state, sizeof(LMState), hipMemcpyHostToDevice, stream));
    // column scalings and camera step cleared, trial camera buffers = copies of the current ones (points-only mode needs valid ones)
    hipLaunchKernelGGL(k_lm_reset<T>, dim3(256), dim3(256), 0, stream, D2p.p, D2p.n, D2c.p, delta_c.p, n, cams[cur].p, cams[1 - cur].p,
                       campre[cur].p, campre[1 - cur].p, C * CAMPRE);
    push_ptrs();
    cur_at_begin = cur;
    if (d_log.n == 0) d_log.alloc(LOG_CAP);
    if (sq_mode()) {
      // points never move in these variants, but the buffer parity flips with every accepted step: both sides equal
      HIPCHK(hipMemcpyAsync(pts[1 - cur].p, pts[cur].p, sizeof(double) * (size_t)N * 3, hipMemcpyDeviceToDevice, stream));
      HIPCHK(hipMemcpyAsync(ptsT[1 - cur].p, ptsT[cur].p, sizeof(T) * (size_t)N * 3, hipMemcpyDeviceToDevice, stream));
      sq_setup();
      if (opts.mode == SBA_MODE_TRANSFORM_SQ) {
        const double ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};     // x0 of pySBA.py:193
        for (int b = 0; b < 2; ++b) HIPCHK(hipMemcpyAsync(theta[b].p, ident, sizeof ident, hipMemcpyHostToDevice, stream));
      }
      launch_sq_linearize(nullptr);
      double g0 = 0;
      sq_read(c0, g0);
      initial_cost = c0;
      h_state->cost = c0;
      HIPCHK(hipMemcpyAsync(d_state.p, h_state, sizeof(LMState), hipMemcpyHostToDevice, stream));
      sync();                      // `ident` above is a stack array
    }
    // no host synchronisation here: everything above is ordered on the stream in front of the first iteration, and the pinned
    // state mirror it was copied from is next written by lm_poll's device-to-host copy on the same stream
    log.clear();
    log_read = 0;
    n_decides = 0;
    poll_clean = false;
    pending_decide = false;
    gmax_cur = gmax_part.p;
    lm_active = true;
    if (!std::isfinite(c0)) { err = "Residuals are not finite in the initial point."; return SBA_ERR_NONFINITE; }
    return SBA_OK;
  }

  int lm_linearize() override {
    if (!lm_active) { err = "sba_lm_begin has not been called"; return SBA_ERR_STATE; }
    poll_clean = false;
    if (!bf3_path()) flush_decide();
    if (sq_mode()) { launch_sq_linearize(d_state.p); return SBA_OK; }
    if (fused()) return SBA_OK;          // k_schur_fused linearises
    if (!lin_pts()) {                    // (f64 one-group rigs: k_schur_sym<LIN> linearises the points)
      prof_begin(KP_LINP);
      launch_linearize_points(d_state.p);
      prof_end(KP_LINP);
    }
    if (h_state->free_cams && !(pairs_fold_u && diag_pairs_bf3())) { prof_begin(KP_LINC); launch_linearize_cams(d_state.p); prof_end(KP_LINC); }
    return SBA_OK;
  }

  int lm_form_reduced(double* E) override { return form_reduced(E, nullptr); }
  int form_reduced(double* E, double* Pk) {
    if (!lm_active) { err = "sba_lm_begin has not been called"; return SBA_ERR_STATE; }
    poll_clean = false;
    if (!bf3_path()) flush_decide();
    if (sq_mode()) {
      if (opts.mode == SBA_MODE_CAMS_ONLY_SQ)
        hipLaunchKernelGGL(k_sq_pack_cams, dim3(64), dim3(256), 0, stream, sq_16.p, C, d_state.p, E);
      return SBA_OK;
    }
    if (h_state->free_cams) {
      prof_begin(KP_SCHUR);
      launch_schur();
      prof_end(KP_SCHUR);
    }
    prof_begin(KP_REDUCE);
    {
      const int fc = (int)h_state->free_cams;
      const bool wide = fused() && fused_wide;
      // (wide: one slab per workgroup with ntw (ntw + 1) / 2 tiles, passed as one "pair")
      // tile slots per pair the grid covers: one pair -> its own tile count (no workgroups that return at once)
      const int ntw = wide_ntw(C);
      const int nt_launch = wide ? ntw * (ntw + 1) / 2 : npairs == 1 ? GROUP_TILES * (GROUP_TILES + 1) / 2 : GROUP_TILES * GROUP_TILES;
      const int np_arg = wide ? 1 : npairs;
      const int nblocks = (fc ? 4 * nt_launch * np_arg + (n + 15) / 16 : 0) + 1;
      hipLaunchKernelGGL(k_build_exchange<T>, dim3(nblocks), dim3(1024), 0, stream, slabs.p, bpart.p, ksplit, pair_ga.p,
                         pair_gb.p, np_arg, U.p, gc.p, cost_part.p, n_lin_parts(), C, fc, E, d_state.p,
                         (fused() || (pairs_fold_u && diag_pairs_bf3())) ? gdpart.p : (const double*)nullptr, Pk, (fused() && fused_wide) ? 3 : (fused() && fused_bf3 && sizeof(T) == 4) ? 1 : (diag_pairs_bf3() ? (offdiag_pairs_bf3() ? 1 : 2) : 0),
                         nt_launch);
    }
    prof_end(KP_REDUCE);
    return SBA_OK;
  }

  // A = S + lam D (n_sys x n_sys, in E) -> chol_sol = A^-1 rhs, chol_info != 0 when A is not positive definite
  // factorisation, both substitutions and the LM epilogue (k_chol_epilogue's work) of a large reduced system
  void launch_chol_big(double* Esys, int n_sys, bool tied) {
    const int npad = cholbig_npad(n_sys), nbr = npad / BB, nbx = (n_sys + BB - 1) / BB;
    if (chol_W.n < (size_t)npad * npad) {
      chol_W.alloc((size_t)npad * npad); chol_Minv.alloc((size_t)nbr * BB * BB); chol_Ld.alloc((size_t)nbr * BB * BB);
      chol_yv.alloc(2 * (size_t)npad);      // k_chol_big_back_all: x, one copy per parity of its epoch, "empty" until stored
      std::vector<long long> empty(2 * (size_t)npad, CHOLBIG_X_EMPTY);
      HIPCHK(hipMemcpyAsync(chol_yv.p, empty.data(), empty.size() * sizeof(long long), hipMemcpyHostToDevice, stream));
      sync();
    }
    if (chol_sol.n < (size_t)n_sys) { chol_sol.alloc(n_sys); chol_info.alloc(1); }
    const size_t lds = (size_t)CHOLBIG_LDS_BLOCKS * CBS * sizeof(double);
    const bool dag = chol_big_dag && !card_shared && chol_big_back_one && nbr <= chol_dag_max_nbr && nbx <= CHOLBIG_MAX_NBX;     // (the per-block back substitution reads the dense copies)
    const bool dag32 = dag && sizeof(T) == 4 && chol_f32;
    if (dag) {
      // factorisation (and k_chol_big_prepare's work) in one launch: a walker workgroup on the diagonal, workgroup = tile behind it,
      // columns handed over through flags (sba_chol_big.hpp).  fp32 engine: on f32 lanes first, the f64 instance behind it only
      // runs when that one refused the system (LMState::chol_retry)
      if (chol_dag_flags.n == 0) { chol_dag_flags.alloc(choldag_nflags(CHOLDAG_MAX_NBR)); chol_dag_flags.zero(stream); }
      if (chol_Mimg.n == 0) chol_Mimg.alloc((size_t)CHOLDAG_MAX_NBR * choldag_img<double>());
      if (chol_debug) { chol_dbg.alloc(8 * (CHOLDAG_MAX_NBR + 1)); chol_dbg.zero(stream); }
      const dim3 grid(1 + nbr * (nbr + 1) / 2);
      if (dag32)
        hipLaunchKernelGGL(k_chol_big_dag<float>, grid, dim3(CHOLBIG_THREADS), (size_t)CHOLBIG_LDS_BLOCKS * CholLay<float>::BS * sizeof(float), stream,
                           Esys, n_sys, d_state.p, D2c.p, reinterpret_cast<float*>(chol_W.p), npad, reinterpret_cast<float*>(chol_Mimg.p),
                           chol_dag_flags.p, ++chol_dag_epoch, chol_info.p, 0, chol_f32_tau, chol_debug ? chol_dbg.p : nullptr);
      hipLaunchKernelGGL(k_chol_big_dag<double>, grid, dim3(CHOLBIG_THREADS), lds, stream, Esys, n_sys, d_state.p, D2c.p, chol_W.p, npad,
                         chol_Mimg.p, chol_dag_flags.p, ++chol_dag_epoch, chol_info.p, dag32 ? 1 : 0, 0.0,
                         (chol_debug && !dag32) ? chol_dbg.p : nullptr);
      if (chol_debug) {
        std::vector<long long> sv(8 * (CHOLDAG_MAX_NBR + 1));
        HIPCHK(hipMemcpyAsync(sv.data(), chol_dbg.p, sv.size() * sizeof(long long), hipMemcpyDeviceToHost, stream));
        sync();
        fprintf(stderr, "[chol_dag%s, the walker, x10 ns: wait for W(c,c-1), W(c,c) | load | panel | downdate | factor+inverse | image stored | flag]\n", dag32 ? " on f32 lanes" : "");
        for (int cc = 0; cc < nbr; ++cc) {
          const long long* v = sv.data() + 8 * cc;
          fprintf(stderr, "  c=%2d at %6lld:", cc, v[4] - sv[0]);
          if (cc > 0) fprintf(stderr, " wait %4lld | load %4lld | panel %4lld | downdate %4lld |", v[1] - sv[8 * (cc - 1) + 7], v[2] - v[1], v[3] - v[2], v[4] - v[3]);
          fprintf(stderr, " factor %4lld | image %4lld | flag %4lld | column %5lld\n", v[5] - v[4], v[6] - v[5], v[7] - v[6], cc > 0 ? v[7] - sv[8 * (cc - 1) + 7] : v[7] - v[0]);
        }
        chol_debug = false;
      }
    } else {
      hipLaunchKernelGGL(k_chol_big_prepare, dim3(nbr * (nbr + 1) / 2), dim3(256), 0, stream, Esys, n_sys, d_state.p, D2c.p,
                         chol_W.p, npad, chol_info.p);
      for (int j = 0; j < nbr; ++j) {
        const int q = nbr - 1 - j;
        hipLaunchKernelGGL(k_chol_big_step, dim3(std::max(1, q * (q + 1) / 2)), dim3(CHOLBIG_THREADS), lds, stream, chol_W.p, npad, j,
                           chol_Minv.p, chol_Ld.p, chol_info.p, d_state.p);
      }
    }
    if (chol_big_back_one && nbx <= CHOLBIG_MAX_NBX) {
      // the whole back substitution in one launch: block row = workgroup, x_b handed over as its own flag; block row 0 runs the epilogue
      hipLaunchKernelGGL(k_chol_big_back_all<T>, dim3(nbx), dim3(256), 0, stream, (const void*)chol_W.p, npad, n_sys, chol_Ld.p, chol_Minv.p,
                         chol_yv.p, ++chol_epoch, chol_sol.p, chol_info.p, d_state.p, dag ? (const void*)chol_Mimg.p : (const void*)nullptr,
                         dag32 ? 1 : 0, Esys, C, D2c.p, ps_lm(), delta_c.p, tied ? tie_map.p : nullptr, tied ? tie_first.p : nullptr);
      return;
    }
    hipLaunchKernelGGL(k_chol_big_back_init, dim3((npad + 255) / 256), dim3(256), 0, stream, chol_W.p, npad, n_sys, chol_Ld.p,
                       chol_yv.p, d_state.p);
    for (int b = nbx - 1; b >= 0; --b)
      hipLaunchKernelGGL(k_chol_big_back, dim3(b + 1), dim3(256), 0, stream, chol_W.p, npad, b, n_sys, chol_Minv.p, chol_yv.p,
                         chol_sol.p, d_state.p);
    hipLaunchKernelGGL(k_chol_epilogue<T>, dim3(1), dim3(1024), 0, stream, Esys, C, n_sys, d_state.p, D2c.p, ps_lm(), delta_c.p,
                       chol_sol.p, chol_info.p, tied ? tie_map.p : nullptr, tied ? tie_first.p : nullptr);
  }

  // scal == nullptr: single rank, the partials are folded inside k_decide and no scalar exchange is needed
  int lm_solve_trial(double* E, double* scal) override {
    if (!lm_active) { err = "sba_lm_begin has not been called"; return SBA_ERR_STATE; }
    poll_clean = false;
    if (opts.mode == SBA_MODE_TRANSFORM_SQ) {
      hipLaunchKernelGGL(k_sq_solve12, dim3(1), dim3(64), 0, stream, sq_16.p, d_state.p, tsets());
      launch_sq_trial();
      return SBA_OK;
    }
    if (h_state->free_cams) {
      prof_begin(KP_CHOL);
      const bool tied = (opts.mode == SBA_MODE_SHARED_INTR);
      const int n_sys = tied ? n_tied : n;
      double* Esys = E;
      if (tied) {   // collapse the camera system onto the tied unknowns
        hipLaunchKernelGGL(k_tie_system, dim3(n_tied), dim3(256), 0, stream, E, n, n_tied, tie_pre_start.p, tie_pre_idx.p,
                           E_tied.p, d_state.p);
        Esys = E_tied.p;
      }
      if (chol_ll && (n_sys > CHOL_LDS_MAX_N || chol_ll_all) && n_sys <= CLL_MAX_NB * CB && C * NCP <= CLL_THREADS) {
        // up to 256 unknowns (23 cameras): left-looking, factor on chip (sba_chol_ll.hpp)
        const int nb = (n_sys + CB - 1) / CB;
        if (chol_work.n < (size_t)nb * (nb + 1) / 2 * CB * CB) chol_work.alloc((size_t)CLL_MAX_NB * (CLL_MAX_NB + 1) / 2 * CB * CB);
        if (chol_debug && chol_dbg.n < 128) { chol_dbg.alloc(128); chol_dbg.zero(stream); }
        // fp32 engine (round 4): the right-looking all-in-LDS kernel on f32 lanes in front of it (the f32 triangle of up to 16 block
        // rows fits the LDS: 20-float rows up to 14 block rows, 17-float rows at 15 and 16); the f64 kernel then only runs when that
        // factorisation refused the system (LMState::chol_retry)
        int only_if_retry = 0;
        if constexpr (sizeof(T) == 4) {
          if (chol_f32 && !chol_ll_all && n_sys > CHOL_LDS_MAX_N && C * NCP <= CHOLB_LDS_THREADS) {
            only_if_retry = 1;
            if (nb <= 14) {
              const size_t lds32 = ((size_t)(nb * (nb + 1) / 2) * CB * 20 + 2 * (size_t)nb * CB) * sizeof(float);
              hipLaunchKernelGGL((k_cholesky_blocked<T, 16, 20>), dim3(1), dim3(CHOLB_LDS_THREADS), lds32, stream, Esys, C, d_state.p, D2c.p,
                                 ps_lm(), delta_c.p, n_sys, tied ? tie_map.p : nullptr, tied ? tie_first.p : nullptr,
                                 (long long*)nullptr, 1, chol_f32_tau);
            } else {
              const size_t lds32 = ((size_t)(nb * (nb + 1) / 2) * CB * 17 + 2 * (size_t)nb * CB) * sizeof(float);
              hipLaunchKernelGGL((k_cholesky_blocked<T, 16, 17>), dim3(1), dim3(CHOLB_LDS_THREADS), lds32, stream, Esys, C, d_state.p, D2c.p,
                                 ps_lm(), delta_c.p, n_sys, tied ? tie_map.p : nullptr, tied ? tie_first.p : nullptr,
                                 (long long*)nullptr, 1, chol_f32_tau);
            }
          }
        }
        hipLaunchKernelGGL(k_cholesky_ll<T>, dim3(1), dim3(CLL_THREADS), CLL_LDS_BYTES, stream, Esys, C, d_state.p, D2c.p,
                           ps_lm(), delta_c.p, n_sys, tied ? tie_map.p : nullptr, tied ? tie_first.p : nullptr, chol_work.p,
                           chol_debug ? chol_dbg.p : nullptr, only_if_retry);
        if (chol_debug) {
          std::vector<long long> st(128);
          HIPCHK(hipMemcpyAsync(st.data(), chol_dbg.p, 128 * sizeof(long long), hipMemcpyDeviceToHost, stream));
          sync();
          if (nb > 6) {
            const long long x0 = st[2 + 2 * 3], y0 = st[3 + 2 * 3];      // start of X_3 / Y_3 (thread 0)
            fprintf(stderr, "[chol_ll step 3, cycles after the step's barrier, per wave] X done:");
            for (int w = 0; w < 8; ++w) fprintf(stderr, " %lld", st[64 + w] - x0);
            fprintf(stderr, " | Y done:");
            for (int w = 0; w < 8; ++w) fprintf(stderr, " %lld", st[72 + w] - y0);
            fprintf(stderr, "\n[chol_ll back substitution, block row 5] x done (per wave, from wave 0's):");
            for (int w = 0; w < 8; ++w) fprintf(stderr, " %lld", st[80 + w] - st[80]);
            fprintf(stderr, " | barrier passed %lld | update done:", st[96] - st[80]);
            for (int w = 0; w < 8; ++w) fprintf(stderr, " %lld", st[88 + w] - st[96]);
            fprintf(stderr, " | barrier passed %lld\n", st[97] - st[96]);
          }
          fprintf(stderr, "[chol_ll stamps, cycles] setup %lld  chol0 %lld |", st[1] - st[0], st[2] - st[1]);
          for (int j = 0; j < nb; ++j) fprintf(stderr, " X%d %lld Y%d %lld |", j, st[3 + 2 * j] - st[2 + 2 * j], j, st[4 + 2 * j] - st[3 + 2 * j]);
          fprintf(stderr, " backsub %lld  epilogue %lld  total %lld\n", st[3 + 2 * nb] - st[2 + 2 * nb], st[4 + 2 * nb] - st[3 + 2 * nb], st[4 + 2 * nb] - st[0]);
          chol_debug = false;
        }
      } else if (n_sys <= CHOL_LDS_MAX_N && !chol_old && C * NCP <= CHOLB_LDS_THREADS) {
        const int nb = (n_sys + CB - 1) / CB;
        const size_t lds = ((size_t)(nb * (nb + 1) / 2) * CBS + 2 * (size_t)nb * CB) * sizeof(double);
        if (chol_debug && chol_dbg.n == 0) { chol_dbg.alloc(64); }
        hipLaunchKernelGGL(k_cholesky_blocked<T>, dim3(1), dim3(CHOLB_LDS_THREADS), lds, stream, Esys, C, d_state.p, D2c.p,
                           ps_lm(), delta_c.p, n_sys, tied ? tie_map.p : nullptr, tied ? tie_first.p : nullptr,
                           chol_debug ? chol_dbg.p : nullptr, (sizeof(T) == 4 && chol_f32) ? 1 : 0, chol_f32_tau);
        if (chol_debug) {
          std::vector<long long> st(64);
          HIPCHK(hipMemcpyAsync(st.data(), chol_dbg.p, 64 * sizeof(long long), hipMemcpyDeviceToHost, stream));
          sync();
          fprintf(stderr, "[chol stamps, cycles] load %lld  chol0 %lld |", st[1] - st[0], st[2] - st[1]);
          for (int j = 0; j < nb; ++j) fprintf(stderr, " B%d %lld C%d %lld |", j, st[3 + 2 * j] - st[2 + 2 * j], j, st[4 + 2 * j] - st[3 + 2 * j]);
          fprintf(stderr, " backsub %lld  epilogue %lld  total %lld\n", st[3 + 2 * nb] - st[2 + 2 * nb], st[4 + 2 * nb] - st[3 + 2 * nb], st[4 + 2 * nb] - st[0]);
          fprintf(stderr, "[chol C0, cycles after B0's barrier, per wave]");
          for (int w = 0; w < CHOLB_LDS_THREADS / 64; ++w) fprintf(stderr, " %lld", st[40 + w] - st[3]);
          fprintf(stderr, "\n");
          chol_debug = false;
        }
      } else if (n_sys <= CHOL_LDS_MAX_N) {
        const size_t lds = (size_t)n_sys * (n_sys + 1) / 2 * sizeof(double);
        hipLaunchKernelGGL((k_cholesky_solve<true, T>), dim3(1), dim3(CHOL_THREADS), lds, stream, Esys, C, d_state.p, D2c.p,
                           ps_lm(), delta_c.p, n_sys, tied ? tie_map.p : nullptr, tied ? tie_first.p : nullptr);
      } else if (n_sys > chol_big_min_n && !chol_old) {
        // 47+ cameras: multi-workgroup right-looking factorisation, one launch per 64-wide block column (sba_chol_big.hpp)
        launch_chol_big(Esys, n_sys, tied);
      } else if (n_sys <= CS_MAX_NB * CB && !chol_old) {
        // 17 .. 46 cameras: one workgroup, left-looking, finished block columns streamed through L2
        const int nb = (n_sys + CB - 1) / CB;
        if (chol_sol.n < (size_t)n_sys) { chol_sol.alloc(n_sys); chol_info.alloc(1); }
        if (chol_work.n < (size_t)nb * (nb + 1) / 2 * CB * CB) chol_work.alloc((size_t)nb * (nb + 1) / 2 * CB * CB);
        // panel (nb blocks) + rhs + as much staging as the 150 KB budget leaves (fewer, larger streaming rounds)
        const size_t fixed = ((size_t)nb * CBS + (size_t)nb * CB) * sizeof(double);
        const int scap = std::min(64, std::max(nb, (int)((150 * 1024 - fixed) / (CBS * sizeof(double)))));
        const size_t lds = fixed + (size_t)scap * CBS * sizeof(double);
        hipLaunchKernelGGL(k_chol_prepare, dim3((n_sys + 255) / 256), dim3(256), 0, stream, Esys, n_sys, d_state.p, D2c.p, chol_sol.p);
        hipLaunchKernelGGL(k_cholesky_stream, dim3(1), dim3(CHOLB_THREADS), lds, stream, Esys, n_sys, chol_work.p, chol_sol.p,
                           chol_info.p, d_state.p, scap, chol_debug ? (chol_dbg.n ? chol_dbg.p : (chol_dbg.alloc(64), chol_dbg.p)) : nullptr);
        if (chol_debug) {
          std::vector<long long> st(8);
          HIPCHK(hipMemcpyAsync(st.data(), chol_dbg.p, 8 * sizeof(long long), hipMemcpyDeviceToHost, stream));
          sync();
          fprintf(stderr, "[chol_stream cycles] update %lld  factor %lld  solve %lld  write-back %lld  back-substitution %lld\n", st[0], st[1], st[2], st[3], st[4]);
          chol_debug = false;
        }
        hipLaunchKernelGGL(k_chol_epilogue<T>, dim3(1), dim3(1024), 0, stream, Esys, C, n_sys, d_state.p, D2c.p, ps_lm(), delta_c.p,
                           chol_sol.p, chol_info.p, tied ? tie_map.p : nullptr, tied ? tie_first.p : nullptr);
      } else {
        hipLaunchKernelGGL((k_cholesky_solve<false, T>), dim3(1), dim3(CHOL_THREADS), 0, stream, Esys, C, d_state.p, D2c.p,
                           ps_lm(), delta_c.p, n_sys, tied ? tie_map.p : nullptr, tied ? tie_first.p : nullptr);
      }
      prof_end(KP_CHOL);
    } else {
      hipLaunchKernelGGL(k_nocam_step, dim3(1), dim3(64), 0, stream, d_state.p, E, n);
      if (N > 0)
        hipLaunchKernelGGL(k_point_factor<T>, dim3((N + 255) / 256), dim3(256), 0, stream, V.p, gp.p, D2p.p, d_state.p, N, pfac.p,
                         has_fixed ? pt_fixed_mask.p : (const unsigned char*)nullptr);
    }
    if (sq_mode()) { launch_sq_trial(); return SBA_OK; }
    prof_begin(KP_BACKSUB);
    launch_backsub_trial();
    prof_end(KP_BACKSUB);
    if (scal)
      hipLaunchKernelGGL(k_trial_scalars, dim3(1), dim3(DECIDE_THREADS), 0, stream, trial_part.p, gmax_rd(), n_trial_parts(), n_lin_parts(), d_state.p, scal);
    return SBA_OK;
  }

  // enqueue the accept / reject / terminate kernel; nothing is read back
  int lm_decide_async(const double* scal_all, int n_ranks) override {
    if (!lm_active) { err = "sba_lm_begin has not been called"; return SBA_ERR_STATE; }
    poll_clean = false;
    ++n_decides;
    if (defer_decide && bf3_path()) {
      // the next k_schur_fused_bf3 takes the decision in its prologue; whatever else needs the decided record first
      // (lm_poll, another linearisation path) enqueues the kernel below through flush_decide()
      pending_decide = true; pend_scal = scal_all; pend_ranks = n_ranks;
      pslot_advance();
      return SBA_OK;
    }
    launch_decide(scal_all, n_ranks);
    pslot_advance();
    return SBA_OK;
  }
  // several camera groups, fp32, indexed producers: the group pairs run on the bf16 pipe (k_schur_diag_bf3 / k_schur_offdiag_bf3)
  bool diag_pairs_bf3() const { return sizeof(T) == 4 && ngroups > 1 && grp_indexed && !fused() && !no_bf3_pairs; }
  bool offdiag_pairs_bf3() const { return diag_pairs_bf3() && !no_bf3_offdiag; }
  // (the name is historical: every Schur kernel that takes the previous step's decision in its prologue -- k_schur_fused_bf3,
  //  k_schur_fused_wide, k_schur_fused_f64)
  bool bf3_path() const { return fused() && (sizeof(T) == 4 ? fused_bf3 : (fused_f64 || fused_wide)) && !sq_mode(); }
  const double* gmax_rd() const { return (bf3_path() && gmax_cur) ? gmax_cur : gmax_part.p; }
  void launch_decide(const double* scal_all, int n_ranks) {
    hipLaunchKernelGGL(k_decide<T>, dim3(1), dim3(DECIDE_THREADS), 0, stream, d_state.p, scal_all, n_ranks, trial_part.p,
                       gmax_rd(), sq_mode() ? nblk_sq : n_trial_parts(), sq_mode() ? nblk_sq : n_lin_parts(), reinterpret_cast<LMLogRow*>(d_log.p), LOG_CAP);
  }
  void flush_decide() {
    if (!pending_decide) return;
    launch_decide(pend_scal, pend_ranks);
    pending_decide = false;
  }

  // read the state back (one sync); returns the scipy status or -1 while the solve is still running
  int lm_poll(int32_t* status_out, int32_t* iterations_out) override {
    if (!lm_active) { err = "sba_lm_begin has not been called"; return SBA_ERR_STATE; }
    flush_decide();
    HIPCHK(hipMemcpyAsync(h_state, d_state.p, sizeof(LMState), hipMemcpyDeviceToHost, stream));
    // the log rows written since the last poll travel with the state: at most one row per accept/reject kernel enqueued so far,
    // into the pinned landing area when they fit (the usual case: a batch of iterations), otherwise by a second copy below
    constexpr size_t ROW_DOUBLES = (sizeof(sba_lm_iter_log) + 7) / 8;
    const int maybe = std::min(n_decides, (int)LOG_CAP) - log_read;
    const bool rows_landed = maybe > 0 && (size_t)maybe * ROW_DOUBLES <= LAND_DOUBLES;
    if (rows_landed)
      HIPCHK(hipMemcpyAsync(h_land, d_log.p + log_read, sizeof(sba_lm_iter_log) * maybe, hipMemcpyDeviceToHost, stream));
    sync_spin();
    HIPCHK(hipGetLastError());
    prof_collect();
    const LMState& s = *h_state;
    if (s.comm_fail) { err = "a peer rank did not reach the exchange in time (sba_ipc; SBA_IPC_TIMEOUT_S, default 5 s): the sharded solve was stopped and the handle can take no further exchange"; lm_active = false; ipc_dead = true; return SBA_ERR_STATE; }
    cur = cur_at_begin ^ (s.cur & 1);
    const int have = std::min(s.iter, LOG_CAP);
    if (have > log_read) {
      log.resize(have);
      if (rows_landed && have - log_read <= maybe)
        std::memcpy(log.data() + log_read, h_land, sizeof(sba_lm_iter_log) * (have - log_read));
      else
        HIPCHK(hipMemcpy(log.data() + log_read, d_log.p + log_read, sizeof(sba_lm_iter_log) * (have - log_read), hipMemcpyDeviceToHost));
      log_read = have;
    }
    poll_clean = true;
    if (status_out) *status_out = s.status;
    if (iterations_out) *iterations_out = s.iter;
    return SBA_OK;
  }

  int lm_decide(const double* scal_all, int n_ranks, int32_t* status_out, int32_t* accepted_out, sba_lm_iter_log* row) override {
    int rc = lm_decide_async(scal_all, n_ranks);
    if (rc) return rc;
    int32_t st = -1, it = 0;
    rc = lm_poll(&st, &it);
    if (rc) return rc;
    if (row && !log.empty()) *row = log.back();
    if (status_out) *status_out = st;
    if (accepted_out) *accepted_out = h_state->accepted;
    return SBA_OK;
  }

  int lm_finish(double* cams_out, double* pts_out, sba_lm_report* rep) override {
    if (!lm_active) { err = "sba_lm_begin has not been called"; return SBA_ERR_STATE; }
    int32_t st = -1, it = 0;
    int rc = poll_clean ? SBA_OK : lm_poll(&st, &it);          // refreshes `cur` and the log (unless the caller just polled)
    if (rc) return rc;
    push_ptrs();                         // table consistent with `cur` for the unconditional launches below
    if (sq_mode()) {
      launch_sq_linearize(nullptr);
      double cost = 0, gmax = 0;
      sq_read(cost, gmax);
      if (opts.mode == SBA_MODE_TRANSFORM_SQ) {
        HIPCHK(hipMemcpyAsync(h_theta, theta[cur].p, sizeof h_theta, hipMemcpyDeviceToHost, stream));
        std::vector<double> X((size_t)N * 3);
        HIPCHK(hipMemcpyAsync(X.data(), pts[cur].p, sizeof(double) * X.size(), hipMemcpyDeviceToHost, stream));
        sync();
        if (pts_out)
          for (int p = 0; p < N; ++p)
            for (int k = 0; k < 3; ++k)
              pts_out[3 * (size_t)p + k] = h_theta[4 * k] * X[3 * (size_t)p] + h_theta[4 * k + 1] * X[3 * (size_t)p + 1] +
                                           h_theta[4 * k + 2] * X[3 * (size_t)p + 2] + h_theta[4 * k + 3];
      } else if (pts_out) {
        HIPCHK(hipMemcpyAsync(pts_out, pts[cur].p, sizeof(double) * (size_t)N * 3, hipMemcpyDeviceToHost, stream));
      }
      if (cams_out) HIPCHK(hipMemcpyAsync(cams_out, cams[cur].p, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
      sync();
      if (rep) {
        const LMState& s = *h_state;
        rep->cost = cost; rep->initial_cost = initial_cost; rep->optimality = gmax; rep->step_norm = s.step_norm;
        rep->lambda = s.lam; rep->nfev = s.nfev; rep->njev = s.njev; rep->iterations = s.iter; rep->accepted = s.n_accepted;
        rep->status = s.status < 0 ? 0 : s.status;
      }
      lm_active = false;
      return SBA_OK;
    }
    // gradient norm at the returned point (scipy reports optimality there, trf.py:546-551)
    launch_linearize_points(nullptr);
    double gmax = 0;
    // the small results land in pinned memory: [gmax partials | cost partials | camera gradient | cameras]
    const int nlp = nblk ? n_linp_blocks() : 0;          // partials of the linearisation launched above
    const size_t need = 2 * (size_t)nlp + 2 * (size_t)n;
    std::vector<double> land_v(need > LAND_DOUBLES ? need : 0);
    double* land = land_v.empty() ? h_land : land_v.data();
    double *gm = land, *cp = land + nlp, *gch = land + 2 * (size_t)nlp, *cams_l = gch + n;
    if (nlp) {
      HIPCHK(hipMemcpyAsync(gm, gmax_part.p, sizeof(double) * nlp, hipMemcpyDeviceToHost, stream));
      HIPCHK(hipMemcpyAsync(cp, cost_part.p, sizeof(double) * nlp, hipMemcpyDeviceToHost, stream));
    }
    if (h_state->free_cams) {
      launch_linearize_cams(nullptr);
      HIPCHK(hipMemcpyAsync(gch, gc.p, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
    } else {
      gc.zero(stream);
      for (int i = 0; i < n; ++i) gch[i] = 0.0;
    }
    if (cams_out) HIPCHK(hipMemcpyAsync(cams_l, cams[cur].p, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
    // the points (1.2 MB at 50k points, into the caller's pageable array: a staged copy that holds the calling thread) go through the
    // copy stream, beside the two gradient kernels enqueued above instead of behind them: the solve has been polled, nothing in
    // flight writes them any more
    if (pts_out) HIPCHK(hipMemcpyAsync(pts_out, pts[cur].p, sizeof(double) * (size_t)N * 3, hipMemcpyDeviceToHost, hres.copy_stream));
    sync_spin();
    if (pts_out) HIPCHK(hipStreamSynchronize(hres.copy_stream));
    HIPCHK(hipGetLastError());
    if (cams_out) std::memcpy(cams_out, cams_l, sizeof(double) * n);
    double cost = 0;
    for (int i = 0; i < nlp; ++i) { gmax = std::max(gmax, gm[i]); cost += cp[i]; }
    if (multi()) {       // whole-job figures: cost and camera gradient are sums over the ranks, the point-gradient maximum a max
      std::vector<double> v(n + 1);
      for (int i = 0; i < n; ++i) v[i] = gch[i];
      v[n] = cost;
      comm_sum(v.data(), n + 1);
      for (int i = 0; i < n; ++i) gch[i] = v[i];
      cost = v[n];
      comm_max(&gmax, 1);
    }
    if (opts.mode == SBA_MODE_SHARED_INTR && (int)h_tie.size() == n) {   // gradient in the tied unknowns
      std::vector<double> gs(n_tied, 0.0);
      for (int i = 0; i < n; ++i) gs[h_tie[i]] += gch[i];
      for (double v : gs) gmax = std::max(gmax, std::fabs(v));
    } else {
      for (int i = 0; i < n; ++i) gmax = std::max(gmax, std::fabs(gch[i]));
    }
    if (rep) {
      const LMState& s = *h_state;
      rep->cost = cost; rep->initial_cost = initial_cost; rep->optimality = gmax; rep->step_norm = s.step_norm;
      rep->lambda = s.lam; rep->nfev = s.nfev; rep->njev = s.njev; rep->iterations = s.iter; rep->accepted = s.n_accepted;
      rep->status = s.status < 0 ? 0 : s.status;
      rep->reserved = s.chol_f64_retries;
    }
    lm_active = false;
    return SBA_OK;
  }

  // the iteration loop of a solve between lm_begin and lm_finish: batches of iterations enqueued back to back, one poll per batch
  int run_loop(const sba_lm_opts* o, int32_t& status, int32_t& iters) {
    int rc = SBA_OK;
    while (status < 0) {
      // a batch of iterations is enqueued back to back; the kernels turn into no-ops once the device-side state says the
      // solve has terminated, and a rejected step skips its re-linearization on the device, not on the host
      // (profiling: one event pair per kernel class, read at every poll.  Two iterations per poll, so that the pairs time the
      //  SECOND one, whose fused kernel carries the previous step's decision in its prologue like every iteration of a normal batch)
      // With a communicator the number of iterations (= collectives) enqueued per poll must not depend on anything rank-local
      // (bf3_path() follows the shard's own visibility density): a rank that enqueues one more all-reduce than its peers after
      // the device-side termination waits for it forever.
      const int pbatch = multi() ? 2 : (bf3_path() && defer_decide ? 2 : 1);
      int batch = prof_on ? pbatch : BATCH;
      if (o->max_iter > 0) batch = std::min(std::max(1, o->max_iter - iters), prof_on ? pbatch : 64);
      for (int b = 0; b < batch; ++b) {
        lm_linearize();
        if (!multi()) {
          lm_form_reduced(E_own.p);
          lm_solve_trial(E_own.p, nullptr);
          lm_decide_async(nullptr, 1);
        } else if (ipc_on) {
          // the same two exchange points through the peer-mapped areas (sba_ipc.hpp): every rank writes its packed system /
          // its 8 trial scalars into its own area, raises a flag, waits (one wave, bounded) for the peers' flags and adds the
          // n_ranks copies in rank order -- no RCCL launch on the iteration's critical path
          const bool fc = h_state->free_cams != 0;
          const unsigned long long s0 = ++ipc_seq[0];
          const size_t sys_off = ipc_L.sys_slot(n, (int)(s0 & 1));
          form_reduced(E_own.p, ipc_mine + sys_off);
          ipc_publish(0, s0, d_state.p);
          ipc_gate(0, s0, d_state.p, 0, 0, nullptr);
          hipLaunchKernelGGL(k_ipc_sum_system, dim3(fc ? std::min(1024, (n * n + 255) / 256) : 1), dim3(256), 0, stream, ipc_ptrs.p, comm_n,
                             sys_off, n, (int)fc, E_own.p, d_state.p);
          const unsigned long long s1 = ++ipc_seq[1];
          const size_t sc_off = ipc_L.scal_slot((int)(s1 & 1));
          lm_solve_trial(E_own.p, ipc_mine + sc_off);
          ipc_publish(1, s1, d_state.p);
          ipc_gate(1, s1, d_state.p, sc_off, NSCAL, sc_all.p);
          lm_decide_async(sc_all.p, comm_n);
        } else {
          // one all-reduce of the packed reduced camera system, one all-gather of 8 scalars per rank; every rank then solves
          // the same system and takes the same decision.  A finished solve turns the kernels into no-ops on the device; the
          // collectives of such a tail still match because every rank enqueues the same number of them.
          if (xpack.n < exch_packed_size(n)) xpack.alloc(exch_packed_size(n));
          form_reduced(E_own.p, xpack.p);
          const bool fc = h_state->free_cams != 0;
          double* xb = fc ? xpack.p : xpack.p + exch_packed_size(n) - 1;       // points-only mode: the cost is all there is
          RCCLCHK(Rccl::get().all_reduce(xb, xb, fc ? exch_packed_size(n) : 1, Rccl::kFloat64, Rccl::kSum, comm, stream));
          hipLaunchKernelGGL(k_unpack_exchange, dim3(fc ? std::min(1024, (n * n + 255) / 256) : 1), dim3(256), 0, stream, xpack.p, n,
                             (int)fc, E_own.p, d_state.p);
          lm_solve_trial(E_own.p, sc_loc.p);
          RCCLCHK(Rccl::get().all_gather(sc_loc.p, sc_all.p, NSCAL, Rccl::kFloat64, comm, stream));
          lm_decide_async(sc_all.p, comm_n);
        }
      }
      rc = lm_poll(&status, &iters);
      if (rc) return rc;
    }
    return SBA_OK;
  }

  // sba_lm_run: the loop alone on a solve begun with sba_lm_begin (bench.py times exactly the iterations with it)
  int lm_run(int32_t* status_out, int32_t* iterations_out) override {
    if (!lm_active) { err = "sba_lm_begin has not been called"; return SBA_ERR_STATE; }
    int32_t status = -1, iters = 0;
    const int rc = run_loop(&opts, status, iters);
    if (rc) return rc;
    if (status_out) *status_out = status;
    if (iterations_out) *iterations_out = iters;
    return SBA_OK;
  }

  int solve(const sba_lm_opts* o, double* cams_out, double* pts_out, sba_lm_report* rep, sba_lm_iter_log* lg, int cap,
            int32_t* rows) override {
    const auto t0 = std::chrono::steady_clock::now();
    static const bool solve_debug = getenv("SBA_SOLVE_DEBUG") != nullptr;      // host-side phase times of one solve on stderr
    auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6; };
    int rc = lm_begin(o);
    if (rc) return rc;
    const double t_begin = since();
    HIPCHK(hipEventRecord(ev0, stream));
    int32_t status = -1, iters = 0;
    rc = run_loop(o, status, iters);
    if (rc) return rc;
    const double t_loop = since();
    HIPCHK(hipEventRecord(ev1, stream));
    HIPCHK(hipEventSynchronize(ev1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev0, ev1));
    rc = lm_finish(cams_out, pts_out, rep);
    if (rc) return rc;
    if (solve_debug)
      fprintf(stderr, "[solve] lm_begin %.0f us | loop (enqueue + polls) %.0f us (device %.0f us) | lm_finish %.0f us | %d iterations\n",
              t_begin, t_loop - t_begin, ms * 1e3, since() - t_loop, (int)iters);
    if (rep) {
      rep->status = status;
      rep->seconds_device = ms * 1e-3;
      rep->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const int nrow = std::min<int>(cap, (int)log.size());
    if (lg) for (int i = 0; i < nrow; ++i) lg[i] = log[i];
    if (rows) *rows = (int32_t)log.size();
    return SBA_OK;
  }

  // ------------------------------------------------------------------ measurement hook
  int time_kernel(const char* name, int reps, double* mean_us) override {
    if (!uploaded) { err = "sba_upload has not been called"; return SBA_ERR_STATE; }
    HIPCHK(hipSetDevice(device));
    const std::string k(name);
    if (reps < 1) reps = 1;
    // make sure everything the kernel reads exists
    if (k == "resjac" && Jc_pm.n != (size_t)M * 2 * NCP) { r_pm.alloc(M); Jc_pm.alloc((size_t)M * 2 * NCP); Jp_pm.alloc((size_t)M * 6); }
    if (k == "schur" || k == "backsub") {
      sba_lm_opts o{}; o.ftol = o.xtol = o.gtol = 0; o.mode = SBA_MODE_FULL;
      if (!lm_active) { int rc = lm_begin(&o); if (rc) return rc; }
      lm_linearize(); lm_form_reduced(E_own.p); lm_solve_trial(E_own.p, scal_own.p);
    } else {
      push_ptrs();
    }
    auto once = [&]() {
      if (k == "residual") launch_residual(nullptr);
      else if (k == "resjac") launch_resjac(r_pm.p);
      else if (k == "linearize_points") launch_linearize_points(nullptr);
      else if (k == "linearize_cams") launch_linearize_cams(nullptr);
      else if (k == "schur") launch_schur();
      else if (k == "backsub") launch_backsub_trial();
      else return false;
      return true;
    };
    if (!once()) { err = "unknown kernel name"; return SBA_ERR_INVALID; }
    sync();
    HIPCHK(hipEventRecord(ev0, stream));
    for (int i = 0; i < reps; ++i) once();
    HIPCHK(hipEventRecord(ev1, stream));
    HIPCHK(hipEventSynchronize(ev1));
    HIPCHK(hipGetLastError());
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev0, ev1));
    *mean_us = (double)ms * 1e3 / reps;
    return SBA_OK;
  }

  // ------------------------------------------------------------------ opt-in extensions
  int set_fixed_points(const uint8_t* mask) override {
    if (!uploaded) { err = "sba_upload has not been called"; return SBA_ERR_STATE; }
    has_fixed = false;
    if (mask) {
      std::vector<unsigned char> m(mask, mask + N);
      for (auto& v : m) { v = v ? 1 : 0; has_fixed = has_fixed || v; }
      if (has_fixed) { pt_fixed_mask.upload(m, stream); sync(); }
    }
    push_ptrs();
    return SBA_OK;
  }
  int set_robust_loss(int loss, double f_scale) override {
    static_assert(SBA_LOSS_HUBER == LOSS_HUBER && SBA_LOSS_SOFT_L1 == LOSS_SOFT_L1 && SBA_LOSS_CAUCHY == LOSS_CAUCHY, "sba_loss values");
    if (loss < SBA_LOSS_LINEAR || loss > SBA_LOSS_CAUCHY) { err = "unknown loss"; return SBA_ERR_INVALID; }
    if (loss != SBA_LOSS_LINEAR && !(f_scale > 0 && std::isfinite(f_scale))) { err = "f_scale must be positive"; return SBA_ERR_INVALID; }
    loss_delta = loss != SBA_LOSS_LINEAR ? f_scale : 0.0;
    loss_kind = loss;
    push_ptrs();
    return SBA_OK;
  }

  // ------------------------------------------------------------------ small accessors of the C ABI
  int set_params_x(const double* x) override {
    if (!uploaded) { err = "not uploaded"; return SBA_ERR_STATE; }
    set_params(x, x + (size_t)C * NCP);
    return SBA_OK;
  }
  int get_params(double* cams_out, double* points_out) override {
    if (!uploaded) { err = "not uploaded"; return SBA_ERR_STATE; }
    if (cams_out) HIPCHK(hipMemcpyAsync(cams_out, cams[cur].p, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
    if (points_out) HIPCHK(hipMemcpyAsync(points_out, pts[cur].p, sizeof(double) * (size_t)N * 3, hipMemcpyDeviceToHost, stream));
    sync();
    return SBA_OK;
  }
  int get_gradient(double* gc_out, double* gp_out) override {
    if (!uploaded) { err = "not uploaded"; return SBA_ERR_STATE; }
    if (gc_out) HIPCHK(hipMemcpyAsync(gc_out, gc.p, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
    if (gp_out) HIPCHK(hipMemcpyAsync(gp_out, gp.p, sizeof(double) * (size_t)N * 3, hipMemcpyDeviceToHost, stream));
    sync();
    return SBA_OK;
  }
  int get_step(double* delta_c_out) override {
    if (!uploaded) { err = "not uploaded"; return SBA_ERR_STATE; }
    HIPCHK(hipMemcpyAsync(delta_c_out, delta_c.p, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
    sync();
    return SBA_OK;
  }
  int get_transform(double* theta12) override { for (int i = 0; i < 12; ++i) theta12[i] = h_theta[i]; return SBA_OK; }
  int get_log(sba_lm_iter_log* lg, int32_t cap, int32_t* rows) override {
    const int nrow = std::min<int>(cap, (int)log.size());
    if (lg) for (int i = 0; i < nrow; ++i) lg[i] = log[i];
    *rows = (int32_t)log.size();
    return SBA_OK;
  }
  int get_kernel_profile(double* total_us, int64_t* count) override {
    for (int k = 0; k < SBA_PROFILE_SLOTS; ++k) { total_us[k] = prof_us[k]; count[k] = prof_cnt[k]; }
    return SBA_OK;
  }
};

// stateless gathered-row calls (sba_rotate / sba_project)
template <typename T>
int rows_call(bool project, int device, int64_t nrows, const double* pts, const double* other, double* out) {
  HIPCHK(hipSetDevice(device));
  if (nrows == 0) return SBA_OK;
  const int64_t ow = project ? NCP : 3, rw = project ? 2 : 3;
  DevBuf<double> d_pts, d_o, d_out;
  d_pts.alloc(nrows * 3); d_o.alloc(nrows * ow); d_out.alloc(nrows * rw);
  HIPCHK(hipMemcpy(d_pts.p, pts, sizeof(double) * nrows * 3, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_o.p, other, sizeof(double) * nrows * ow, hipMemcpyHostToDevice));
  const int g = (int)((nrows + 255) / 256);
  if (project) hipLaunchKernelGGL(k_project_rows<T>, dim3(g), dim3(256), 0, 0, d_pts.p, d_o.p, d_out.p, nrows);
  else hipLaunchKernelGGL(k_rotate_rows<T>, dim3(g), dim3(256), 0, 0, d_pts.p, d_o.p, d_out.p, nrows);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, d_out.p, sizeof(double) * nrows * rw, hipMemcpyDeviceToHost));
  return SBA_OK;
}



inline EngineBase* make_engine(int dtype) {
  if (dtype == SBA_F32) return new Engine<float>();
  return new Engine<double>();
}
inline int rows_call_dtype(int dtype, bool project, int device, int64_t n, const double* pts, const double* other, double* out) {
  return dtype == SBA_F32 ? rows_call<float>(project, device, n, pts, other, out) : rows_call<double>(project, device, n, pts, other, out);
}

}  // namespace SBA_NS
