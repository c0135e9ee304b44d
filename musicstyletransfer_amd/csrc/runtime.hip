// runtime.hip — library plumbing: error string, hipGraph capture helpers, events, and a
// hardware layout self-test that pins the MFMA / ds_read_tr16_b64 lane maps the kernels rely on.
#include <stdarg.h>
#include <string.h>
#include "common.hpp"

namespace mst {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mst

using namespace mst;

extern "C" int mst_version(void) { return 100; }
extern "C" const char* mst_last_error(void) { return g_err; }

extern "C" int mst_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return MST_ERR_LAUNCH;
  }
  return n;
}

#define HIP_TRY(expr)                                             \
  do {                                                            \
    hipError_t e__ = (expr);                                      \
    if (e__ != hipSuccess) {                                      \
      set_error("%s: %s", #expr, hipGetErrorString(e__));         \
      return MST_ERR_LAUNCH;                                      \
    }                                                             \
  } while (0)

extern "C" int mst_graph_begin(mst_stream_t stream) {
  HIP_TRY(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return MST_OK;
}
extern "C" int mst_graph_end(mst_stream_t stream, void** graph_exec_out) {
  MST_CHECK_ARG(graph_exec_out != nullptr, "mst_graph_end: null output");
  hipGraph_t g = nullptr;
  HIP_TRY(hipStreamEndCapture((hipStream_t)stream, &g));
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) {
    set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
    return MST_ERR_LAUNCH;
  }
  *graph_exec_out = (void*)ge;
  return MST_OK;
}
extern "C" int mst_graph_launch(void* graph_exec, mst_stream_t stream) {
  HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return MST_OK;
}
extern "C" int mst_graph_destroy(void* graph_exec) {
  if (graph_exec) HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return MST_OK;
}

extern "C" int mst_event_create(void** ev_out) {
  MST_CHECK_ARG(ev_out != nullptr, "mst_event_create: null output");
  hipEvent_t ev;
  HIP_TRY(hipEventCreate(&ev));
  *ev_out = (void*)ev;
  return MST_OK;
}
extern "C" int mst_event_record(void* ev, mst_stream_t stream) {
  HIP_TRY(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
  return MST_OK;
}
extern "C" int mst_event_sync(void* ev) {
  HIP_TRY(hipEventSynchronize((hipEvent_t)ev));
  return MST_OK;
}
extern "C" int mst_event_elapsed_ms(void* a, void* b, float* ms_out) {
  HIP_TRY(hipEventElapsedTime(ms_out, (hipEvent_t)a, (hipEvent_t)b));
  return MST_OK;
}
extern "C" int mst_event_destroy(void* ev) {
  if (ev) HIP_TRY(hipEventDestroy((hipEvent_t)ev));
  return MST_OK;
}

// ---------------------------------------------------------------------------------------------
// Layout self-test. One wave, exact small-integer data (asymmetric operands), four checks:
//  [0] mfma_f32_16x16x32_bf16: A[row l&15][k 8(l>>4)+j], B[k 8(l>>4)+j][col l&15],
//      D: col = l&15, row = 4(l>>4)+r
//  [1] mfma_f32_32x32x16_bf16: A[row l&31][k 8(l>>5)+j], B[k][col l&31],
//      D: col = l&31, row = (r&3) + 8(r>>2) + 4(l>>5)
//  [2] ds_read_tr16_b64: lane i of a 16-lane group receives column i of the 4 rows whose
//      addresses lanes 4q+p supply (row q, columns 4p..4p+3)
//  [3] accumulator tile reused as the A operand of the next 32x32x16 MFMA computes X^T·B with
//      element j of lane-half h being row 16s + 8(j>>2) + 4h + (j&3) of X
// ---------------------------------------------------------------------------------------------
__global__ void selftest_kernel(int32_t* flags) {
  __shared__ __bf16 tile[32 * 32];
  const int l = threadIdx.x;
  auto Aval = [](int i, int k) { return (float)(((i * 3 + k * 5) % 7) - 3); };
  auto Bval = [](int k, int j) { return (float)(((k * 2 + j * 7) % 5) - 2); };

  // [0] 16x16x32
  {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = (__bf16)Aval(l & 15, 8 * (l >> 4) + j);
      b[j] = (__bf16)Bval(8 * (l >> 4) + j, l & 15);
    }
    f32x4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d, 0, 0, 0);
    int ok = 1;
    for (int r = 0; r < 4; ++r) {
      int row = 4 * (l >> 4) + r, col = l & 15;
      float ref = 0;
      for (int k = 0; k < 32; ++k) ref += Aval(row, k) * Bval(k, col);
      ok &= (d[r] == ref);
    }
    ok = __all(ok);
    if (l == 0) flags[0] = ok;
  }
  // [1] 32x32x16
  f32x16 X;
  {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = (__bf16)Aval(l & 31, 8 * (l >> 5) + j);
      b[j] = (__bf16)Bval(8 * (l >> 5) + j, l & 31);
    }
    f32x16 d;
    for (int r = 0; r < 16; ++r) d[r] = 0;
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d, 0, 0, 0);
    int ok = 1;
    for (int r = 0; r < 16; ++r) {
      int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
      float ref = 0;
      for (int k = 0; k < 16; ++k) ref += Aval(row, k) * Bval(k, col);
      ok &= (d[r] == ref);
    }
    ok = __all(ok);
    if (l == 0) flags[1] = ok;
    X = d;  // X[row][col] = sum_k A[row][k] B[k][col], |X| <= 16*3*2 = 96: exact in bf16? (<=256 ok)
  }
  // [2] transposed LDS read: 32x32 tile with value row*32+col (exact in bf16 up to 256 only → use
  //     row*8 + (col&7) + 64*(col>>3)... keep it simple: value = (row*5 + col*3) % 251, exact)
  auto Tval = [](int r, int c) { return (float)((r * 37 + c * 11) % 251); };
  for (int i = l; i < 32 * 32; i += 64) tile[i] = (__bf16)Tval(i / 32, i % 32);
  __syncthreads();
  {
    // block rows r0 = 4*(l>>5) + 8 (arbitrary, distinct per half), cols c0 = 16*((l>>4)&1)
    int g_i = l & 15, q = g_i >> 2, p = g_i & 3;
    int r0 = 8 + 4 * (l >> 5), c0 = 16 * ((l >> 4) & 1);
    const __bf16* addr = &tile[(r0 + q) * 32 + c0 + 4 * p];
    i16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) i16x4*)(uintptr_t)(addr));
    int ok = 1;
    for (int e = 0; e < 4; ++e) {
      uint16_t bits = (uint16_t)v[e];
      float got = bits_to_f32<__bf16>(bits);
      ok &= (got == Tval(r0 + e, c0 + g_i));
    }
    ok = __all(ok);
    if (l == 0) flags[2] = ok;
  }
  // [3] Y[c][n] = sum_r X[r][c] * Bv[r][n]  with X regs as the A operand (two k-steps of 16 rows)
  {
    auto B2 = [](int r, int n) { return (float)(((r * 3 + n * 2) % 3) - 1); };
    f32x16 y;
    for (int r = 0; r < 16; ++r) y[r] = 0;
    for (int s = 0; s < 2; ++s) {
      bf16x8 a, b;
      for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)X[8 * s + j];
        int xr = 16 * s + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
        b[j] = (__bf16)B2(xr, l & 31);
      }
      y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, y, 0, 0, 0);
    }
    int ok = 1;
    for (int r = 0; r < 16; ++r) {
      int c = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), n = l & 31;
      float ref = 0;
      for (int xr = 0; xr < 32; ++xr) {
        float x = 0;
        for (int k = 0; k < 16; ++k) x += Aval(xr, k) * Bval(k, c);
        ref += x * B2(xr, n);
      }
      ok &= (y[r] == ref);
    }
    ok = __all(ok);
    if (l == 0) flags[3] = ok;
  }
}

extern "C" int mst_selftest(int32_t* out_flags_device, mst_stream_t stream) {
  MST_CHECK_ARG(out_flags_device != nullptr, "mst_selftest: null output");
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_flags_device);
  MST_CHECK_LAUNCH("selftest_kernel");
  return MST_OK;
}
