// resource-usage probe: instantiates only the large-system Cholesky kernels (seconds to compile instead of minutes)
#define SBA_NCP 11
#include "../../lasercalib_amd/csrc/sba_common.hpp"
#include "../../lasercalib_amd/csrc/sba_lm_kernels.hpp"
#include "../../lasercalib_amd/csrc/sba_chol_blocked.hpp"
#include "../../lasercalib_amd/csrc/sba_chol_big.hpp"
template __global__ void SBA_NS::k_chol_big_dag<double>(const double*, int, SBA_NS::LMState*, double*, double*, int, double*, unsigned*, unsigned, int*, int, double, long long*);
template __global__ void SBA_NS::k_chol_big_dag<float>(const double*, int, SBA_NS::LMState*, double*, float*, int, float*, unsigned*, unsigned, int*, int, float, long long*);
