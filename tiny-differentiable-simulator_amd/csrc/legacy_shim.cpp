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
//
// Compiled with -DTDS_SHIM_ABI2 the same file builds  cudalib_<name>.so  with the reference's NEWER
// generated-library ABI instead (loader src/utils/cuda/cuda_library.hpp:38-70, cuda_model.hpp:14-25,
// cuda_function.hpp:4-20,78-140; emitter src/utils/cuda/cuda_codegen.hpp:32-231):
//     void model_info(const char *const **names, int *count);          // -> {"cuda_model_<name>"}
//     CudaFunctionMetaData <model>_forward_zero_meta();                // {output_dim, local_input_dim,
//                                                                      //  global_input_dim, accumulated_output}
//     void <model>_forward_zero_allocate(int);  void <model>_forward_zero_deallocate();
//     bool <model>_forward_zero_send_local(int N, const Float *);      // fprintf(stderr) + false on error
//     bool <model>_forward_zero_send_global(const Float *);            // global_input_dim == 0: no-op, true
//     void <model>_forward_zero(int N, int num_blocks, int num_threads_per_block, Float *output);
// (<model>_jacobian is absent, which CudaFunction tolerates: is_available() == false.)
// The two ABIs reuse symbol names with different signatures, hence two libraries.
//
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

#ifdef TDS_SHIM_ABI2
struct CudaFunctionMetaData {
  int output_dim;
  int local_input_dim;
  int global_input_dim;
  bool accumulated_output;
};
#else
struct CudaFunctionMetaData {
  int output_dim;
  int input_dim;
  int global_dim;
};
#endif

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

#ifdef TDS_SHIM_ABI2
#define TDS_STR2(x) #x
#define TDS_STR(x) TDS_STR2(x)
void model_info(const char *const **names, int *count) {
  static const char *const g_names[] = {"cuda_model_" TDS_STR(TDS_SHIM_MODEL)};
  *names = g_names;
  *count = 1;
}

CudaFunctionMetaData SHIM_FN(_forward_zero_meta)(void) {
  CudaFunctionMetaData d;
  d.output_dim = shim_model()->output_dim;
  d.local_input_dim = shim_model()->input_dim;
  d.global_input_dim = 0;
  d.accumulated_output = false;
  return d;
}
#else
CudaFunctionMetaData SHIM_FN(_forward_zero_meta)(void) {
  CudaFunctionMetaData d;
  d.output_dim = shim_model()->output_dim;
  d.input_dim = shim_model()->input_dim;
  d.global_dim = 0;
  return d;
}
#endif

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

#ifdef TDS_SHIM_ABI2
bool SHIM_FN(_forward_zero_send_local)(int num_total_threads, const Float *input) {
  if (!g_sim || num_total_threads > g_capacity) {
    fprintf(stderr, "tds_hip shim: send_local(%d) without a matching allocate(%d)\n", num_total_threads, g_capacity);
    return false;
  }
  if (tds_hip_send_local(g_sim, num_total_threads, input) != TDS_OK) {
    fprintf(stderr, "tds_hip shim: send_local failed: %s\n", tds_hip_last_error());
    return false;
  }
  return true;
}

bool SHIM_FN(_forward_zero_send_global)(const Float *input) {
  (void)input;  // global_input_dim == 0
  return true;
}

void SHIM_FN(_forward_zero)(int num_total_threads, int num_blocks, int num_threads_per_block, Float *output) {
  (void)num_blocks;
  (void)num_threads_per_block;
  if (!g_sim || num_total_threads > g_capacity) {
    fprintf(stderr, "tds_hip shim: forward_zero(%d) without a matching allocate(%d)\n", num_total_threads, g_capacity);
    exit(1);
  }
  int rc = tds_hip_forward_zero_fetch(g_sim, num_total_threads, output);
  if (rc != TDS_OK) {
    fprintf(stderr, "tds_hip shim: forward_zero failed: %s\n", tds_hip_last_error());
    exit(rc);
  }
}
#else
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
#endif

}  // extern "C"
