/* A host WITHOUT an interpreter: loads a step plan (automl_amd/plan.py) through the network-level C ABI
 * (include/edet_net.h), runs the inference pass and -- if the plan has one -- training steps, and dumps the named
 * buffers for tests/test_plan_gpu.py to compare with what the Python host computed.  C99, links libedet_hip.so and the HIP
 * runtime only.   usage: edet_host PLAN OUTDIR [graph] [steps]  */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "../../include/edet_hip.h"
#include "../../include/edet_net.h"

#define CHECK(call)                                                              \
  do {                                                                           \
    if ((call) != 0) {                                                           \
      fprintf(stderr, "%s failed: %s\n", #call, edet_last_error());             \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

static int dump(edet_net_t* net, const char* name, const char* dir, const char* suffix) {
  void* p = NULL;
  size_t n = 0;
  if (edet_net_buffer(net, name, &p, &n) != 0) return 1;
  void* host = malloc(n ? n : 1);
  if (hipMemcpy(host, p, n, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  char path[1024];
  snprintf(path, sizeof(path), "%s/%s%s.bin", dir, name, suffix);
  for (char* c = path + strlen(dir) + 1; *c; ++c)
    if (*c == ':' || *c == '/') *c = '_';
  FILE* f = fopen(path, "wb");
  if (!f) return 1;
  fwrite(host, 1, n, f);
  fclose(f);
  free(host);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s PLAN OUTDIR [graph] [steps]\n", argv[0]);
    return 2;
  }
  const int graph = argc > 3 && strcmp(argv[3], "graph") == 0;
  const int steps = argc > 4 ? atoi(argv[4]) : 1;
  edet_net_t* net = NULL;
  hipStream_t stream;
  if (hipStreamCreate(&stream) != hipSuccess) return 1;
  CHECK(edet_create(argv[1], &net));
  CHECK(edet_net_use_graph(net, graph));
  int64_t min_level = 0, max_level = 0, batch = 0;
  CHECK(edet_net_property(net, "min_level", &min_level));
  CHECK(edet_net_property(net, "max_level", &max_level));
  CHECK(edet_net_property(net, "batch", &batch));
  /* the inference pass three times (eager, captured, replayed with graph): the outputs of the last one are dumped */
  for (int i = 0; i < 3; ++i) CHECK(edet_forward(net, stream));
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  char name[64];
  for (int64_t l = min_level; l <= max_level; ++l) {
    snprintf(name, sizeof(name), "cls_outputs_%d", (int)l);
    if (dump(net, name, argv[2], "")) return 1;
    snprintf(name, sizeof(name), "box_outputs_%d", (int)l);
    if (dump(net, name, argv[2], "")) return 1;
  }
  if (edet_net_has_program(net, "detect")) {      /* raw uint8 images -> detections (EfficientDetModel.call) */
    for (int i = 0; i < 3; ++i) CHECK(edet_detect(net, stream));
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    const char* det[] = {"detections.boxes", "detections.scores", "detections.classes", "detections.valid_len"};
    for (int k = 0; k < 4; ++k)
      if (dump(net, det[k], argv[2], "")) return 1;
  }
  if (edet_net_has_program(net, "train_step")) {
    for (int s = 0; s < steps; ++s) {
      CHECK(edet_train_step(net, 0.02f, 0.9f, stream));
      if (hipStreamSynchronize(stream) != hipSuccess) return 1;
      char suffix[32];
      snprintf(suffix, sizeof(suffix), ".step%d", s);
      const char* state[] = {"params", "ema", "velocity", "bn_state", "loss_sums"};
      for (int k = 0; k < 5; ++k)
        if (dump(net, state[k], argv[2], suffix)) return 1;
    }
  }
  printf("edet_host: batch %d, levels %d..%d, %d buffers, graph %d, ok\n", (int)batch, (int)min_level, (int)max_level,
         edet_net_num_buffers(net), graph);
  CHECK(edet_destroy(net));
  return 0;
}
