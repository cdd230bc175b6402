/* Network-level C ABI of the MI355X-native EfficientDet path: a whole inference pass / training step behind one call,
 * for a host without a Python interpreter.  Same library (libedet_hip.so), same error convention (0 = ok, otherwise
 * edet_last_error()) as the operator-level ABI of edet_hip.h.
 *
 * What the reference interface is:   efficientdet/tf2/efficientdet_keras.py:790-799  EfficientDetNet.__init__ (model_name,
 *   config) + build            -> edet_create;      :893-915 EfficientDetNet.call(inputs, training) -> edet_forward;
 *   efficientdet/tf2/train_lib.py:606-684 EfficientDetNetTrain.train_step -> edet_train_step;
 *   efficientdet/tf2/anchors.py:117-165 Anchors.__init__/_generate_boxes -> edet_anchors;
 *   efficientdet/tf2/train.py:184-198 (the distribution strategy's gradient all-reduce) -> edet_dp_init.
 *
 * How it works: the launches of a step are a pure function of (config, batch, image size, storage type).  The Python
 * host (automl_amd/plan.py: record_network) runs the step once with a recorder on the operator-level ABI and writes a
 * PLAN: device buffers (sizes, initial contents of the persistent ones: variables, optimizer slots, moving statistics,
 * tables), named handles, and per program the list of edet_* calls with every device pointer as (buffer, offset), the
 * fork / join of the two head chains as event operations, and the place of the gradient exchange.  edet_create loads a
 * plan (hipMalloc + upload + pointer relocation); edet_forward / edet_train_step issue its calls -- eagerly the first
 * time (one-time kernel attribute setup), then, with edet_net_use_graph(net, 1), as ONE captured hipGraph per program.
 * A compiled host therefore needs this header, libedet_hip.so and a plan file -- no interpreter, no torch.
 *
 * Thread safety: one edet_net_t per host thread (the operator-level ABI keeps per-stream deferred-reduction state).  */
#ifndef EDET_NET_H_
#define EDET_NET_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct edet_net edet_net_t;

/* Loads a plan: allocates its buffers on the current HIP device, uploads the initial contents, relocates the pointers.
 * The network starts in the state the plan was recorded from (variables, optimizer slots, moving statistics, and the
 * recorded inputs in "images" / the target buffers).  */
int edet_create(const char* plan_path, edet_net_t** net_out);
int edet_destroy(edet_net_t* net);

/* Named device buffers of the network: "images" [B,H,W,3] in the storage type; "cls_outputs_<L>" / "box_outputs_<L>"
 * [B,h,w,ld] per pyramid level (the first `channels` elements of a pixel are the logits; properties
 * "<name>.channels/.ld/.height/.width/.elem_bytes"); training: "cls_targets_<L>" int32 [B,h,w,A], "box_targets_<L>" fp32
 * [B,h,w,4A], "mean_num_positives" fp32 [B]; state: "params", "ema", "velocity" (flat fp32 arenas), "bn_state" (moving
 * statistics), "loss_sums" fp32[4] (class, box, L2, -: sums of the last step), "hyper" fp32[4]; "drop_mask:<block>"
 * fp32 [B,C] stochastic-depth scales floor(p + u) / p that the host refreshes per step (utils.drop_connect).  */
int edet_net_buffer(edet_net_t* net, const char* name, void** device_ptr, size_t* bytes);
int edet_net_num_buffers(edet_net_t* net);
const char* edet_net_buffer_name(edet_net_t* net, int index);
/* Integer properties: "batch", "height", "width", "min_level", "max_level", "num_classes", "num_anchors",
 * "storage_elem_bytes", "num_train_elems", "<buffer>.channels" ...  */
int edet_net_property(edet_net_t* net, const char* name, int64_t* value);
/* 1 if the plan holds the program ("forward", "train_step"), else 0 */
int edet_net_has_program(edet_net_t* net, const char* program);

/* Synchronous copies between host memory and device buffers (hipMemcpy on the library's HIP runtime), for hosts that do
 * not link the HIP runtime themselves; they wait for the device first.  */
int edet_copy_to_host(void* host, const void* device, size_t bytes);
int edet_copy_to_device(void* device, const void* host, size_t bytes);

/* 1: programs are captured into a hipGraph at their second run and replayed afterwards; 0 (default): eager launches */
int edet_net_use_graph(edet_net_t* net, int on);

/* Inference pass over the contents of "images" (EfficientDetNet.call(inputs, training=False)): fills the
 * "cls_outputs_<L>" / "box_outputs_<L>" buffers.  stream: a hipStream_t (NULL = the default stream).  */
int edet_forward(edet_net_t* net, void* stream);

/* Detection on raw images (efficientdet_keras.EfficientDetModel.call, tf2/efficientdet_keras.py:920-1000, with the reference's
 * defaults pre_mode='infer', post_mode='global'): "raw_images" uint8 [B, raw_height, raw_width, 3] -> normalise / resize /
 * pad (:920-951) -> the network -> pre_nms + global NMS + clip + rescale to raw pixels (tf2/postprocess.py:375-406) ->
 * "detections.boxes" fp32 [B, M, 4] (ymin, xmin, ymax, xmax), "detections.scores" fp32 [B, M], "detections.classes" fp32
 * [B, M] (1-based), "detections.valid_len" int32 [B]; M = property "max_output_size".  Plans recorded with
 * record_network(..., detect_raw_hw=(H, W)) hold the program.  */
int edet_detect(edet_net_t* net, void* stream);

/* One training step over "images" and the target buffers (EfficientDetNetTrain.train_step: forward with batch
 * statistics, focal + Huber loss, backward, L2, per-tensor and global-norm clip, [gradient exchange], SGD momentum + EMA).
 * learning_rate / ema_decay are this step's values of the schedule (train_lib.py:37-173, :193-197; ema_decay 0 = the
 * plan was recorded without a moving average).  */
int edet_train_step(edet_net_t* net, float learning_rate, float ema_decay, void* stream);

/* Data parallelism (one process per GPU): `fn` is called where the reference's optimizer all-reduces the clipped
 * gradients (train_lib.py:675-683), with the flat fp32 gradient arena, to be summed IN PLACE over the replicas on
 * `stream` (e.g. ncclAllReduce(buf, buf, count, ncclFloat, ncclSum, comm, stream) -- RCCL captures into the step's
 * graph).  fn = NULL: single replica (the default).  */
typedef int (*edet_allreduce_fn)(void* ctx, float* buf, size_t count, void* stream);
int edet_dp_init(edet_net_t* net, edet_allreduce_fn fn, void* ctx);

/* Anchor boxes of tf2/anchors.py Anchors (float64 arithmetic, cast to float32; level-major, then y, x, then
 * a = octave * num_aspects + aspect): boxes_out [count][4] = (ymin, xmin, ymax, xmax) on the HOST.  aspect_ratios: the
 * scalar form (w/h ratio).  boxes_out may be NULL (count only); capacity in boxes.  */
int edet_anchors(int min_level, int max_level, int num_scales, const double* aspect_ratios, int num_aspects,
                 double anchor_scale, int image_height, int image_width, float* boxes_out, int64_t capacity,
                 int64_t* count);

#ifdef __cplusplus
}
#endif
#endif  /* EDET_NET_H_ */
