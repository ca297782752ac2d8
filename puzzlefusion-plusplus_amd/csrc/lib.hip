// Library-level entry points of libpfpp_hip.so: version, error text, device query.
#include <stdarg.h>
#include <string.h>

#include "pfpp_common.h"

namespace pfpp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return PFPP_EHIP;
  }
  return PFPP_OK;
}

// arithmetic mode of the attention kernels for launches issued by THIS thread (pfpp_set_attention_mode): like the error text, state
// of the calling thread, not of the process - two host threads driving two streams never see each other's setting
static thread_local int g_attn_mode = -1;
int attn_mode() { return g_attn_mode; }

}  // namespace pfpp

extern "C" int pfpp_set_attention_mode(int mode) {
  if (mode < -1 || mode > 2) {
    pfpp::set_error("pfpp_set_attention_mode: mode must be -1 (defaults), 0 (exact fp32), 1 (split-f16) or 2 (single-pass fp16)");
    return PFPP_EINVAL;
  }
  pfpp::g_attn_mode = mode;
  return PFPP_OK;
}

extern "C" int pfpp_get_attention_mode(void) { return pfpp::g_attn_mode; }

extern "C" int pfpp_version(void) { return PFPP_ABI_VERSION; }

// What this binary was compiled with (pfpp_hip/build.py passes the switch AND its attestation macro together in the flags every
// translation unit gets; a build that bypasses it reports "unattested" and pfpp_hip._lib.load() refuses the library):
//   fma_mix_insts=off    no v_fma_mix* : every fp32 -> fp16 conversion of a hi / lo split is one v_cvt of the rounded value
//   packed_fp32_ops=off  no v_pk_{add,mul,fma}_f32 : round 5's wrong farthest point next to a co-running GEMM
#ifdef PFPP_ATTEST_NO_MIX
#define PFPP_BI_MIX "off"
#else
#define PFPP_BI_MIX "unattested"
#endif
#ifdef PFPP_ATTEST_NO_PK
#define PFPP_BI_PK "off"
#else
#define PFPP_BI_PK "unattested"
#endif
#define PFPP_STR2(x) #x
#define PFPP_STR(x) PFPP_STR2(x)
extern "C" const char* pfpp_build_info(void) {
  return "abi=" PFPP_STR(PFPP_ABI_VERSION) ";arch=gfx950;fma_mix_insts=" PFPP_BI_MIX ";packed_fp32_ops=" PFPP_BI_PK ";chain_prio=" PFPP_STR(PFPP_CHAIN_PRIO);
}

extern "C" const char* pfpp_last_error(void) { return pfpp::g_err; }

extern "C" int64_t pfpp_abi_sizeof(const char* name) {
  if (!name) return -1;
#define PFPP_SZ(n) if (!strcmp(name, #n)) return (int64_t)sizeof(pfpp_##n);
  PFPP_SZ(sample_level) PFPP_SZ(gemm_args) PFPP_SZ(planes) PFPP_SZ(slab_job) PFPP_SZ(gemm_planes_args) PFPP_SZ(sa_train_args)
  PFPP_SZ(gemm_grad_args) PFPP_SZ(tlayer_params) PFPP_SZ(tlayer_grads) PFPP_SZ(tlayer_adamw) PFPP_SZ(tlayers_args) PFPP_SZ(pw)
  PFPP_SZ(elayer_params) PFPP_SZ(tlayers_eval_args) PFPP_SZ(head_params) PFPP_SZ(head_grads) PFPP_SZ(reblock_job) PFPP_SZ(dw_job)
#undef PFPP_SZ
  return -1;
}

extern "C" int pfpp_device_cu_count(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
  return prop.multiProcessorCount;
}
