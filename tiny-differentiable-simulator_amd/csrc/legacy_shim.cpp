// legacy_shim.cpp — the reference's generated-library ABI on top of libtds_hip.so.
//
// Built once per model as  cuda_model_<name>.so  exporting exactly the symbols the reference's
// CudaModel<double>("cuda_model_" + env_name()) looks up with dlsym
// (reference: examples/ars/ars_train_policy_cuda.cpp:220-229, 345-359, 507; emitter
//  src/utils/cuda_codegen.hpp:146-262):
//     CudaFunctionMetaData <model>_forward_zero_meta();
//     void <model>_forward_zero_allocate(int num_total_threads);
//     void <model>_forward_zero_deallocate();
//     void <model>_forward_zero(int num_total_threads, int num_blocks, int num_threads_per_block,
//                               Float *output, const Float *input);       // Float = double
// Semantics kept: file-scope state (one instance per process), blocking call, input =
// [global_dim globals | N * input_dim], output = N * output_dim, errors -> fprintf(stderr) + exit.
// num_blocks / num_threads_per_block are accepted and ignored: the launch shape of the MI355X
// kernel (wave-group per environment) is not the one-thread-per-environment shape of the
// generated CUDA kernel.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tds_hip.h"

#ifndef TDS_SHIM_MODEL
#error "compile with -DTDS_SHIM_MODEL=<name> -DTDS_SHIM_BLOB=\"<file>.inc\""
#endif

#define TDS_CAT2(a, b) a##b
#define TDS_CAT(a, b) TDS_CAT2(a, b)
#define SHIM_FN(suffix) TDS_CAT(TDS_CAT(cuda_model_, TDS_SHIM_MODEL), suffix)

typedef double Float;

struct CudaFunctionMetaData {
  int output_dim;
  int input_dim;
  int global_dim;
};

static const unsigned char g_blob[] = {
#include TDS_SHIM_BLOB
};

static tds_hip_sim_t *g_sim = NULL;
static int g_capacity = 0;

static const tds_model_t *shim_model(void) {
  static tds_model_t m;
  static int init = 0;
  if (!init) {
    if (sizeof(g_blob) != sizeof(tds_model_t)) {
      fprintf(stderr, "tds_hip shim: embedded model blob has the wrong size\n");
      exit(1);
    }
    memcpy(&m, g_blob, sizeof(m));
    init = 1;
  }
  return &m;
}

extern "C" {

CudaFunctionMetaData SHIM_FN(_forward_zero_meta)(void) {
  CudaFunctionMetaData d;
  d.output_dim = shim_model()->output_dim;
  d.input_dim = shim_model()->input_dim;
  d.global_dim = 0;
  return d;
}

void SHIM_FN(_forward_zero_allocate)(int num_total_threads) {
  if (g_sim) {
    tds_hip_destroy(g_sim);
    g_sim = NULL;
  }
  int dev = 0;
  const char *e = getenv("TDS_HIP_DEVICE");
  if (e) dev = atoi(e);
  int rc = tds_hip_create(shim_model(), num_total_threads, dev, TDS_DTYPE_F64, &g_sim);
  if (rc != TDS_OK) {  // reference: allocation failure -> message + exit(status)
    fprintf(stderr, "tds_hip shim: allocate(%d) failed: %s\n", num_total_threads, tds_hip_last_error());
    exit(rc);
  }
  g_capacity = num_total_threads;
}

void SHIM_FN(_forward_zero_deallocate)(void) {
  tds_hip_destroy(g_sim);
  g_sim = NULL;
  g_capacity = 0;
}

void SHIM_FN(_forward_zero)(int num_total_threads, int num_blocks, int num_threads_per_block, Float *output,
                            const Float *input) {
  (void)num_blocks;
  (void)num_threads_per_block;
  if (!g_sim || num_total_threads > g_capacity) {
    fprintf(stderr, "tds_hip shim: forward_zero(%d) without a matching allocate(%d)\n", num_total_threads, g_capacity);
    exit(1);
  }
  int rc = tds_hip_forward_zero_host(g_sim, num_total_threads, input /* global_dim == 0 */, output);
  if (rc != TDS_OK) {
    fprintf(stderr, "tds_hip shim: forward_zero failed: %s\n", tds_hip_last_error());
    exit(rc);
  }
}

}  // extern "C"
