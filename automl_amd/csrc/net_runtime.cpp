// Network-level C ABI (include/edet_net.h): loads a step plan written by automl_amd/plan.py and replays its programs --
// the launches of EfficientDetNet.call (efficientdet/tf2/efficientdet_keras.py:893-915) and of
// EfficientDetNetTrain.train_step (efficientdet/tf2/train_lib.py:606-684) -- through the operator-level ABI of this
// library, eagerly or as one captured hipGraph per program.  Host code only: no kernels in this file.
//
// A plan holds no code and no host pointers: every device pointer is (buffer, offset), every structure / array argument is
// a byte blob with relocation entries, every stream an index (0 = the caller's stream), every fork / join an event
// operation.  File layout: automl_amd/plan.py (docstring).
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "../../include/edet_net.h"

namespace {

union PlanSlot {
  int64_t i;
  double f;
  void* p;
};

#include "plan_stubs.inc"

constexpr uint32_t NULL_BUF = 0xffffffffu;
enum OpKind : uint8_t { OP_CALL = 0, OP_EVENT_RECORD = 1, OP_STREAM_WAIT = 2, OP_ALLREDUCE = 3 };
enum ArgType : uint8_t { A_INT = 0, A_DOUBLE = 1, A_DEVPTR = 2, A_STREAM = 3, A_BLOB = 4, A_NULL = 5 };

struct Op {
  uint8_t kind;
  int fn;                       // OP_CALL: index into kPlanFnNames
  std::vector<PlanSlot> args;   // OP_CALL: decoded slots (stream slots are patched per run)
  std::vector<int> stream_args; // OP_CALL: (slot index << 8) | stream index
  uint32_t a = 0, b = 0;        // event / stream indices
  void* ptr = nullptr;          // OP_ALLREDUCE
  uint64_t count = 0;
};

struct Program {
  std::string name;
  std::vector<Op> ops;
  int runs = 0;
  hipGraphExec_t exec = nullptr;
  hipGraph_t graph = nullptr;
  hipStream_t captured_on = nullptr;
};

struct Named {
  uint32_t buf;
  uint64_t off, bytes;
};

}  // namespace

struct edet_net {
  std::vector<void*> bufs;
  std::vector<uint64_t> buf_bytes;
  std::vector<std::string> name_list;
  std::map<std::string, Named> names;
  std::map<std::string, int64_t> props;
  std::vector<Program> programs;
  std::vector<hipStream_t> streams;      // [0] unused (the caller's stream)
  std::vector<hipEvent_t> events;
  std::vector<std::vector<unsigned char>*> blobs;
  bool use_graph = false;
  edet_allreduce_fn allreduce = nullptr;
  void* allreduce_ctx = nullptr;
};

namespace {

struct Reader {
  FILE* f;
  bool ok = true;
  template <class T> T get() {
    T v{};
    if (fread(&v, sizeof(T), 1, f) != 1) ok = false;
    return v;
  }
  std::string str() {
    const uint16_t n = get<uint16_t>();
    std::string s(n, '\0');
    if (n && fread(&s[0], 1, n, f) != n) ok = false;
    return s;
  }
};

void free_net(edet_net* net) {
  if (!net) return;
  for (auto& p : net->programs) {
    if (p.exec) (void)hipGraphExecDestroy(p.exec);
    if (p.graph) (void)hipGraphDestroy(p.graph);
  }
  for (size_t i = 1; i < net->streams.size(); ++i)
    if (net->streams[i]) (void)hipStreamDestroy(net->streams[i]);
  for (auto e : net->events)
    if (e) (void)hipEventDestroy(e);
  for (auto b : net->bufs)
    if (b) (void)hipFree(b);
  for (auto b : net->blobs) delete b;
  delete net;
}

#define NET_CHECK(cond, ...)          \
  do {                                \
    if (!(cond)) {                    \
      edet_set_error(__VA_ARGS__);    \
      free_net(net);                  \
      if (f) fclose(f);               \
      return -1;                      \
    }                                 \
  } while (0)

int find_fn(const std::string& name) {
  const int n = (int)(sizeof(kPlanFnNames) / sizeof(kPlanFnNames[0]));
  for (int i = 0; i < n; ++i)
    if (name == kPlanFnNames[i]) return i;
  return -1;
}

Program* find_program(edet_net* net, const char* name) {
  for (auto& p : net->programs)
    if (p.name == name) return &p;
  return nullptr;
}

// Issues the operations of a program on `main` (stream index 0) and the network's own streams.
int issue(edet_net* net, Program& prog, hipStream_t main) {
  std::vector<PlanSlot> slots;
  for (Op& op : prog.ops) {
    switch (op.kind) {
      case OP_CALL: {
        slots = op.args;
        for (int sa : op.stream_args) {
          const int idx = sa & 0xff;
          slots[sa >> 8].p = idx == 0 ? (void*)main : (void*)net->streams[idx];
        }
        const int rc = plan_dispatch(op.fn, slots.data());
        if (rc != 0) return rc;      // the entry point has set the error text
        break;
      }
      case OP_EVENT_RECORD: {
        const hipError_t e = hipEventRecord(net->events[op.a], op.b == 0 ? main : net->streams[op.b]);
        EDET_CHECK(e == hipSuccess, "plan '%s': hipEventRecord: %s", prog.name.c_str(), hipGetErrorString(e));
        break;
      }
      case OP_STREAM_WAIT: {
        const hipError_t e = hipStreamWaitEvent(op.a == 0 ? main : net->streams[op.a], net->events[op.b], 0);
        EDET_CHECK(e == hipSuccess, "plan '%s': hipStreamWaitEvent: %s", prog.name.c_str(), hipGetErrorString(e));
        break;
      }
      case OP_ALLREDUCE: {
        if (net->allreduce) {
          const int rc = net->allreduce(net->allreduce_ctx, (float*)op.ptr, (size_t)op.count,
                                        op.a == 0 ? (void*)main : (void*)net->streams[op.a]);
          EDET_CHECK(rc == 0, "plan '%s': the gradient all-reduce callback returned %d", prog.name.c_str(), rc);
        }
        break;
      }
      default:
        EDET_CHECK(false, "plan '%s': unknown operation %d", prog.name.c_str(), (int)op.kind);
    }
  }
  return 0;
}

int run_program(edet_net* net, const char* name, void* stream) {
  EDET_CHECK(net, "edet_net: null network");
  Program* prog = find_program(net, name);
  EDET_CHECK(prog, "edet_net: the plan holds no program '%s'", name);
  hipStream_t main = reinterpret_cast<hipStream_t>(stream);
  // first run eager (one-time kernel attribute setup of the library is not capturable); with use_graph the second run is
  // captured -- the side streams join the capture through the recorded fork / join events -- and replayed from then on
  if (net->use_graph && prog->runs >= 1) {
    if (prog->exec && prog->captured_on != main) {      // a graph is tied to nothing, but keep one exec per stream simple
      (void)hipGraphExecDestroy(prog->exec);
      (void)hipGraphDestroy(prog->graph);
      prog->exec = nullptr;
      prog->graph = nullptr;
    }
    if (!prog->exec) {
      EDET_CHECK(main != nullptr, "edet_net: graph mode needs a non-default stream (stream capture)");
      hipError_t e = hipStreamBeginCapture(main, hipStreamCaptureModeThreadLocal);
      EDET_CHECK(e == hipSuccess, "edet_net: hipStreamBeginCapture: %s", hipGetErrorString(e));
      const int rc = issue(net, *prog, main);
      hipGraph_t g = nullptr;
      e = hipStreamEndCapture(main, &g);
      if (rc != 0) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
      }
      EDET_CHECK(e == hipSuccess && g, "edet_net: hipStreamEndCapture: %s", hipGetErrorString(e));
      hipGraphExec_t x = nullptr;
      e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
      if (e != hipSuccess) {
        (void)hipGraphDestroy(g);
        EDET_CHECK(false, "edet_net: hipGraphInstantiate: %s", hipGetErrorString(e));
      }
      prog->graph = g;
      prog->exec = x;
      prog->captured_on = main;
    }
    const hipError_t e = hipGraphLaunch(prog->exec, main);
    EDET_CHECK(e == hipSuccess, "edet_net: hipGraphLaunch: %s", hipGetErrorString(e));
    ++prog->runs;
    return 0;
  }
  const int rc = issue(net, *prog, main);
  if (rc == 0) ++prog->runs;
  return rc;
}

}  // namespace

extern "C" int edet_create(const char* plan_path, edet_net_t** net_out) {
  edet_net* net = nullptr;
  FILE* f = nullptr;
  NET_CHECK(plan_path && net_out, "edet_create: null argument");
  *net_out = nullptr;
  f = fopen(plan_path, "rb");
  NET_CHECK(f, "edet_create: cannot open %s", plan_path);
  net = new edet_net();
  Reader r{f};
  char magic[8];
  NET_CHECK(fread(magic, 1, 8, f) == 8 && memcmp(magic, "EDETPLAN", 8) == 0, "edet_create: %s is not a plan file", plan_path);
  const uint32_t version = r.get<uint32_t>(), nbuf = r.get<uint32_t>(), nnames = r.get<uint32_t>(),
                 nstreams = r.get<uint32_t>(), nevents = r.get<uint32_t>(), nprog = r.get<uint32_t>(),
                 nfn = r.get<uint32_t>(), ndevreloc = r.get<uint32_t>();
  NET_CHECK(r.ok && version == 1, "edet_create: plan version %u (this library reads version 1)", version);
  NET_CHECK(nstreams >= 1 && nstreams <= 255, "edet_create: bad stream count %u", nstreams);
  std::vector<int> fn_map(nfn);
  for (uint32_t i = 0; i < nfn; ++i) {
    const std::string name = r.str();
    fn_map[i] = find_fn(name);
    NET_CHECK(fn_map[i] >= 0, "edet_create: the plan calls %s, which this library does not export", name.c_str());
  }
  std::vector<uint64_t> init_at(nbuf);
  net->bufs.assign(nbuf, nullptr);
  net->buf_bytes.assign(nbuf, 0);
  for (uint32_t i = 0; i < nbuf; ++i) {
    net->buf_bytes[i] = r.get<uint64_t>();
    init_at[i] = r.get<uint64_t>();
  }
  NET_CHECK(r.ok, "edet_create: truncated plan (buffer table)");
  for (uint32_t i = 0; i < nbuf; ++i) {
    const hipError_t e = hipMalloc(&net->bufs[i], net->buf_bytes[i] ? net->buf_bytes[i] : 1);
    NET_CHECK(e == hipSuccess, "edet_create: hipMalloc(%llu bytes) for buffer %u: %s",
              (unsigned long long)net->buf_bytes[i], i, hipGetErrorString(e));
    if (!init_at[i]) {
      const hipError_t m = hipMemset(net->bufs[i], 0, net->buf_bytes[i]);
      NET_CHECK(m == hipSuccess, "edet_create: hipMemset: %s", hipGetErrorString(m));
    }
  }
  auto dev = [&](uint32_t b, uint64_t off) -> void* {
    return b == NULL_BUF ? nullptr : (void*)((char*)net->bufs[b] + off);
  };
  for (uint32_t i = 0; i < nnames; ++i) {
    const std::string name = r.str();
    Named n;
    n.buf = r.get<uint32_t>();
    n.off = r.get<uint64_t>();
    n.bytes = r.get<uint64_t>();
    NET_CHECK(r.ok && (n.buf == NULL_BUF || (n.buf < nbuf && n.off + n.bytes <= net->buf_bytes[n.buf])),
              "edet_create: bad named buffer '%s'", name.c_str());
    if (n.buf == NULL_BUF) {
      net->props[name] = (int64_t)n.off;
    } else {
      net->names[name] = n;
      net->name_list.push_back(name);
    }
  }
  struct DevReloc { uint32_t buf; uint64_t at; uint32_t tbuf; uint64_t toff; };
  std::vector<DevReloc> devreloc(ndevreloc);
  for (auto& d : devreloc) {
    d.buf = r.get<uint32_t>(); d.at = r.get<uint64_t>(); d.tbuf = r.get<uint32_t>(); d.toff = r.get<uint64_t>();
    NET_CHECK(r.ok && d.buf < nbuf && d.tbuf < nbuf && d.at + 8 <= net->buf_bytes[d.buf] && d.toff <= net->buf_bytes[d.tbuf],
              "edet_create: bad device relocation");
  }
  net->streams.assign(nstreams, nullptr);
  for (uint32_t i = 1; i < nstreams; ++i) {
    const hipError_t e = hipStreamCreateWithFlags(&net->streams[i], hipStreamNonBlocking);
    NET_CHECK(e == hipSuccess, "edet_create: hipStreamCreate: %s", hipGetErrorString(e));
  }
  net->events.assign(nevents, nullptr);
  for (uint32_t i = 0; i < nevents; ++i) {
    const hipError_t e = hipEventCreateWithFlags(&net->events[i], hipEventDisableTiming);
    NET_CHECK(e == hipSuccess, "edet_create: hipEventCreate: %s", hipGetErrorString(e));
  }
  net->programs.resize(nprog);
  for (uint32_t pi = 0; pi < nprog; ++pi) {
    Program& prog = net->programs[pi];
    prog.name = r.str();
    const uint32_t nops = r.get<uint32_t>();
    NET_CHECK(r.ok, "edet_create: truncated plan (program header)");
    prog.ops.resize(nops);
    for (uint32_t oi = 0; oi < nops; ++oi) {
      Op& op = prog.ops[oi];
      op.kind = r.get<uint8_t>();
      if (op.kind == OP_CALL) {
        const uint16_t fid = r.get<uint16_t>();
        const uint8_t nargs = r.get<uint8_t>();
        NET_CHECK(r.ok && fid < nfn, "edet_create: bad entry-point index in program '%s'", prog.name.c_str());
        op.fn = fn_map[fid];
        NET_CHECK(nargs == kPlanFnArgs[op.fn], "edet_create: %s takes %d arguments, the plan passes %d (plan from another "
                  "version of the library?)", kPlanFnNames[op.fn], kPlanFnArgs[op.fn], (int)nargs);
        op.args.resize(nargs);
        for (int k = 0; k < nargs; ++k) {
          const uint8_t t = r.get<uint8_t>();
          PlanSlot s;
          s.i = 0;
          if (t == A_INT) {
            s.i = r.get<int64_t>();
          } else if (t == A_DOUBLE) {
            s.f = r.get<double>();
          } else if (t == A_DEVPTR) {
            const uint32_t b = r.get<uint32_t>();
            const uint64_t off = r.get<uint64_t>();
            NET_CHECK(r.ok && (b == NULL_BUF || (b < nbuf && off <= net->buf_bytes[b])), "edet_create: bad device pointer");
            s.p = dev(b, off);
          } else if (t == A_STREAM) {
            const uint32_t idx = r.get<uint32_t>();
            NET_CHECK(r.ok && idx < nstreams, "edet_create: bad stream index");
            op.stream_args.push_back((k << 8) | (int)idx);
          } else if (t == A_BLOB) {
            const uint32_t nbytes = r.get<uint32_t>();
            NET_CHECK(r.ok && nbytes <= (1u << 20), "edet_create: bad argument blob");
            auto* blob = new std::vector<unsigned char>((nbytes + 15) / 8 * 8);
            net->blobs.push_back(blob);
            if (nbytes) NET_CHECK(fread(blob->data(), 1, nbytes, f) == nbytes, "edet_create: truncated plan (blob)");
            const uint16_t nreloc = r.get<uint16_t>();
            for (int q = 0; q < nreloc; ++q) {
              const uint32_t at = r.get<uint32_t>(), b = r.get<uint32_t>();
              const uint64_t off = r.get<uint64_t>();
              NET_CHECK(r.ok && at + 8 <= nbytes && b < nbuf && off <= net->buf_bytes[b], "edet_create: bad blob relocation");
              void* p = dev(b, off);
              memcpy(blob->data() + at, &p, 8);
            }
            s.p = blob->data();
          } else {
            NET_CHECK(t == A_NULL, "edet_create: unknown argument type %d", (int)t);
          }
          op.args[k] = s;
        }
      } else if (op.kind == OP_EVENT_RECORD || op.kind == OP_STREAM_WAIT) {
        op.a = r.get<uint32_t>();
        op.b = r.get<uint32_t>();
        const uint32_t ev = op.kind == OP_EVENT_RECORD ? op.a : op.b, st = op.kind == OP_EVENT_RECORD ? op.b : op.a;
        NET_CHECK(r.ok && ev < nevents && st < nstreams, "edet_create: bad event operation");
      } else if (op.kind == OP_ALLREDUCE) {
        const uint32_t b = r.get<uint32_t>();
        const uint64_t off = r.get<uint64_t>();
        op.count = r.get<uint64_t>();
        op.a = r.get<uint32_t>();
        NET_CHECK(r.ok && b < nbuf && off + 4 * op.count <= net->buf_bytes[b] && op.a < nstreams, "edet_create: bad all-reduce operation");
        op.ptr = dev(b, off);
      } else {
        NET_CHECK(false, "edet_create: unknown operation %d", (int)op.kind);
      }
    }
    NET_CHECK(r.ok, "edet_create: truncated plan (program '%s')", prog.name.c_str());
  }
  // initial contents, through one pinned staging buffer
  {
    const size_t CH = 64u << 20;
    void* stage = nullptr;
    NET_CHECK(hipHostMalloc(&stage, CH, hipHostMallocDefault) == hipSuccess, "edet_create: hipHostMalloc failed");
    bool good = true;
    for (uint32_t i = 0; i < nbuf && good; ++i) {
      if (!init_at[i]) continue;
      good = fseek(f, (long)init_at[i], SEEK_SET) == 0;
      for (uint64_t done = 0; good && done < net->buf_bytes[i]; done += CH) {
        const size_t n = (size_t)(net->buf_bytes[i] - done < CH ? net->buf_bytes[i] - done : CH);
        good = fread(stage, 1, n, f) == n && hipMemcpy((char*)net->bufs[i] + done, stage, n, hipMemcpyHostToDevice) == hipSuccess;
      }
    }
    (void)hipHostFree(stage);
    NET_CHECK(good, "edet_create: could not read / upload the initial contents");
  }
  for (const auto& d : devreloc) {
    void* p = dev(d.tbuf, d.toff);
    NET_CHECK(hipMemcpy((char*)net->bufs[d.buf] + d.at, &p, 8, hipMemcpyHostToDevice) == hipSuccess,
              "edet_create: device relocation upload failed");
  }
  NET_CHECK(hipDeviceSynchronize() == hipSuccess, "edet_create: hipDeviceSynchronize failed");
  fclose(f);
  *net_out = net;
  return 0;
}

extern "C" int edet_destroy(edet_net_t* net) {
  if (net) (void)hipDeviceSynchronize();
  free_net(net);
  return 0;
}

extern "C" int edet_net_buffer(edet_net_t* net, const char* name, void** device_ptr, size_t* bytes) {
  EDET_CHECK(net && name, "edet_net_buffer: null argument");
  auto it = net->names.find(name);
  EDET_CHECK(it != net->names.end(), "edet_net_buffer: the plan names no buffer '%s'", name);
  if (device_ptr) *device_ptr = (char*)net->bufs[it->second.buf] + it->second.off;
  if (bytes) *bytes = (size_t)it->second.bytes;
  return 0;
}

extern "C" int edet_net_num_buffers(edet_net_t* net) { return net ? (int)net->name_list.size() : 0; }

extern "C" const char* edet_net_buffer_name(edet_net_t* net, int index) {
  if (!net || index < 0 || index >= (int)net->name_list.size()) return nullptr;
  return net->name_list[index].c_str();
}

extern "C" int edet_net_property(edet_net_t* net, const char* name, int64_t* value) {
  EDET_CHECK(net && name && value, "edet_net_property: null argument");
  auto it = net->props.find(name);
  EDET_CHECK(it != net->props.end(), "edet_net_property: the plan holds no property '%s'", name);
  *value = it->second;
  return 0;
}

extern "C" int edet_net_has_program(edet_net_t* net, const char* program) {
  return net && program && find_program(net, program) ? 1 : 0;
}

extern "C" int edet_copy_to_host(void* host, const void* device, size_t bytes) {
  EDET_CHECK((host && device) || bytes == 0, "edet_copy_to_host: null pointer");
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess && bytes) e = hipMemcpy(host, device, bytes, hipMemcpyDeviceToHost);
  EDET_CHECK(e == hipSuccess, "edet_copy_to_host: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int edet_copy_to_device(void* device, const void* host, size_t bytes) {
  EDET_CHECK((host && device) || bytes == 0, "edet_copy_to_device: null pointer");
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess && bytes) e = hipMemcpy(device, host, bytes, hipMemcpyHostToDevice);
  EDET_CHECK(e == hipSuccess, "edet_copy_to_device: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int edet_net_use_graph(edet_net_t* net, int on) {
  EDET_CHECK(net, "edet_net_use_graph: null network");
  net->use_graph = on != 0;
  return 0;
}

extern "C" int edet_forward(edet_net_t* net, void* stream) { return run_program(net, "forward", stream); }

extern "C" int edet_detect(edet_net_t* net, void* stream) { return run_program(net, "detect", stream); }

extern "C" int edet_train_step(edet_net_t* net, float learning_rate, float ema_decay, void* stream) {
  EDET_CHECK(net, "edet_train_step: null network");
  auto it = net->names.find("hyper");
  EDET_CHECK(it != net->names.end() && it->second.bytes >= 8, "edet_train_step: the plan names no 'hyper' buffer");
  // per-step scalars of the schedule: a stream-ordered copy from pageable memory (staged by the runtime before it
  // returns), in front of -- never inside -- the replayed graph, as Engine.set_hyper does
  const float h[2] = {learning_rate, ema_decay};
  const hipError_t e = hipMemcpyAsync((char*)net->bufs[it->second.buf] + it->second.off, h, sizeof(h), hipMemcpyHostToDevice,
                                      reinterpret_cast<hipStream_t>(stream));
  EDET_CHECK(e == hipSuccess, "edet_train_step: hipMemcpyAsync: %s", hipGetErrorString(e));
  return run_program(net, "train_step", stream);
}

extern "C" int edet_dp_init(edet_net_t* net, edet_allreduce_fn fn, void* ctx) {
  EDET_CHECK(net, "edet_dp_init: null network");
  Program* p = find_program(net, "train_step");
  EDET_CHECK(p, "edet_dp_init: the plan holds no training step");
  net->allreduce = fn;
  net->allreduce_ctx = ctx;
  if (p->exec) {      // the captured step was recorded without / with another exchange: capture again at the next run
    (void)hipGraphExecDestroy(p->exec);
    (void)hipGraphDestroy(p->graph);
    p->exec = nullptr;
    p->graph = nullptr;
  }
  return 0;
}

// tf2/anchors.py:117-165 (Anchors._generate_configs / _generate_boxes) with utils.get_feat_sizes (utils.py:497-526); float64
// arithmetic in the reference's order of operations, cast to float32 at the end.
extern "C" int edet_anchors(int min_level, int max_level, int num_scales, const double* aspect_ratios, int num_aspects,
                            double anchor_scale, int image_height, int image_width, float* boxes_out, int64_t capacity,
                            int64_t* count) {
  EDET_CHECK(min_level >= 0 && max_level >= min_level && max_level < 16 && num_scales > 0 && aspect_ratios && num_aspects > 0 &&
             image_height > 0 && image_width > 0 && count, "edet_anchors: bad arguments");
  int fh[17], fw[17];
  fh[0] = image_height;
  fw[0] = image_width;
  for (int l = 1; l <= max_level; ++l) {
    fh[l] = (fh[l - 1] - 1) / 2 + 1;
    fw[l] = (fw[l - 1] - 1) / 2 + 1;
  }
  int64_t n = 0;
  for (int level = min_level; level <= max_level; ++level) {
    const double sy = (double)fh[0] / (double)fh[level], sx = (double)fw[0] / (double)fw[level];
    // np.arange(stride / 2, image_size, stride): ceil((stop - start) / step) samples start + i * step
    const int64_t ny = (int64_t)ceil(((double)image_height - sy / 2) / sy), nx = (int64_t)ceil(((double)image_width - sx / 2) / sx);
    const int A = num_scales * num_aspects;
    if (boxes_out) {
      EDET_CHECK(n + ny * nx * A <= capacity, "edet_anchors: capacity %lld boxes is too small", (long long)capacity);
      for (int octave = 0; octave < num_scales; ++octave) {
        for (int ai = 0; ai < num_aspects; ++ai) {
          const double octave_scale = (double)octave / (double)num_scales;
          const double base_x = anchor_scale * sx * pow(2.0, octave_scale);
          const double base_y = anchor_scale * sy * pow(2.0, octave_scale);
          const double ax = sqrt(aspect_ratios[ai]), ay = 1.0 / ax;
          const double half_x = base_x * ax / 2.0, half_y = base_y * ay / 2.0;
          const int a = octave * num_aspects + ai;
          for (int64_t y = 0; y < ny; ++y) {
            const double yv = sy / 2 + (double)y * sy;
            for (int64_t x = 0; x < nx; ++x) {
              const double xv = sx / 2 + (double)x * sx;
              float* o = boxes_out + 4 * (n + (y * nx + x) * A + a);
              o[0] = (float)(yv - half_y);
              o[1] = (float)(xv - half_x);
              o[2] = (float)(yv + half_y);
              o[3] = (float)(xv + half_x);
            }
          }
        }
      }
    }
    n += ny * nx * A;
  }
  *count = n;
  return 0;
}
