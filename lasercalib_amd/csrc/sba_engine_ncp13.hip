// sba_engine_ncp13.hip -- the engine and all kernels for the 13-parameter camera model ([rvec, t, f, k1, k2, p1, p2, cx, cy]: radial + tangential, BASELINE config 5).
#define SBA_NCP 13
#include "sba_engine.hpp"

sba_host::EngineBase* sba_make_engine_ncp13(int dtype) { return SBA_NS::make_engine(dtype); }
int sba_rows_call_ncp13(int dtype, bool project, int device, int64_t n, const double* pts, const double* other, double* out) {
  return SBA_NS::rows_call_dtype(dtype, project, device, n, pts, other, out);
}
