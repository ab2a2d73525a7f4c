// sba_engine_ncp11.hip -- the engine and all kernels for the 11-parameter camera model ([rvec, t, f, k1, k2, cx, cy]: the reference's, pySBA.py:31-35).
#define SBA_NCP 11
#include "sba_engine.hpp"

sba_host::EngineBase* sba_make_engine_ncp11(int dtype) { return SBA_NS::make_engine(dtype); }
int sba_rows_call_ncp11(int dtype, bool project, int device, int64_t n, const double* pts, const double* other, double* out) {
  return SBA_NS::rows_call_dtype(dtype, project, device, n, pts, other, out);
}
