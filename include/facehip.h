/* facehip.h — C ABI of libfacehip.so: the MI355X (gfx950) implementation of the
 * detect -> align -> embed -> compare path of cucibala/FaceRecognizeOnnx.
 *
 * This is the drop-in boundary: every entry point names the reference interface it replaces
 * (paths relative to the reference root).  Plain pointers and sizes only; no C++ or torch types.
 * Host-pointer entry points reproduce the reference's batch-1 class API; the *_dev entry
 * points take device pointers (HBM-resident inputs/outputs) and a hipStream_t passed as
 * void*, and are asynchronous on that stream unless stated otherwise.
 *
 * Error convention: functions returning int give >= 0 on success and a negative fh_status on
 * failure; fh_last_error() holds the message of the calling thread's last failure.  Nothing
 * throws across this boundary.  Handles own all device memory; one handle = one device
 * (the device current when it was created) and calls on one handle must not overlap.
 */
#ifndef FACEHIP_H_
#define FACEHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FH_API __attribute__((visibility("default")))

enum fh_status { FH_OK = 0, FH_ERR_ARG = -1, FH_ERR_MODEL = -2, FH_ERR_DEVICE = -3, FH_ERR_STATE = -4 };

/* POD mirror of `struct FaceBox` (src/face_detector.h:8-12): cv::Rect box, float score,
 * cv::Point2f landmarks[5] (left eye, right eye, nose, left mouth, right mouth).  60 bytes. */
typedef struct fh_face {
    int32_t x, y, w, h;
    float score;
    float lm[10];
} fh_face;

typedef struct fh_det fh_det;         /* FaceDetector   (src/face_detector.h:14-43)   */
typedef struct fh_rec fh_rec;         /* FaceRecognizer (src/face_recognizer.h:9-38)  */
typedef struct fh_gallery fh_gallery; /* 1:N extension of compareFaces                */

FH_API const char* fh_version(void);
FH_API const char* fh_last_error(void);
/* Selects the HIP device for subsequently created handles; returns the device count or < 0. */
FH_API int fh_init(int device);

/* ---- host-only introspection (no GPU needed): parse + plan an .onnx, write a text summary. */
FH_API int fh_plan_describe(const char* onnx_path, int default_h, int default_w, char* buf, int cap);
/* The reader's own view of an .onnx file (what loadModel hands to the planner, src/face_detector.cpp:20-90 path), one canonical
 * text line per graph input / output / initializer (name, dtype, dims, element count, fp64 sum, first and last value) / node
 * (op, inputs, outputs, attributes sorted by name).  Exists so that the wire-format reader can be checked against an independent
 * protobuf decoder (tests/test_onnx_pin.py).  Returns the text length (truncated to cap - 1) or < 0. */
FH_API int fh_onnx_dump(const char* onnx_path, char* buf, int cap);

/* ---- FaceDetector ------------------------------------------------------------------------
 * fh_det_create   <- FaceDetector::FaceDetector + loadModel   (src/face_detector.cpp:5-12,20-90)
 *                    NULL on failure (the reference returns false).
 * fh_det_detect   <- FaceDetector::detect                    (src/face_detector.cpp:139-222)
 *                    host BGR u8 image (cv::Mat data/rows/cols/step); returns the number of
 *                    faces written to out (score-descending, <= max_out); 0 for the
 *                    reference's empty-result cases (null/empty image, bad size, unexpected
 *                    output layout).                                                        */
FH_API fh_det* fh_det_create(const char* onnx_path);
FH_API void fh_det_destroy(fh_det* d);
FH_API int fh_det_input_size(const fh_det* d, int* width, int* height);
FH_API int fh_det_num_anchors(const fh_det* d);
FH_API double fh_det_macs_per_frame(const fh_det* d);
FH_API double fh_det_act_bytes_per_frame(const fh_det* d);
FH_API int fh_det_detect(fh_det* d, const uint8_t* bgr, int rows, int cols, int step, float score_thr, float nms_thr,
                         fh_face* out, int max_out);
/* n frames of identical size resident in HBM.  d_out: [n][max_per_frame] fh_face, d_counts: [n]
 * (total survivors per frame; entries beyond max_per_frame are not stored). */
FH_API int fh_det_detect_batch_dev(fh_det* d, const uint8_t* d_frames, int n, int rows, int cols, int step,
                                   long long frame_stride, float score_thr, float nms_thr, fh_face* d_out,
                                   int max_per_frame, int* d_counts, void* stream);
/* Stage hooks for parity tests: run preprocess + network only / read a raw network output
 * (device pointer to [n][rows][cols] fp32, valid until the next call on the handle). */
FH_API int fh_det_run_network_dev(fh_det* d, const uint8_t* d_frames, int n, int rows, int cols, int step,
                                  long long frame_stride, void* stream);
FH_API int fh_det_num_outputs(const fh_det* d);
FH_API const float* fh_det_output_dev(fh_det* d, int index, int* rows, int* cols);
FH_API const float* fh_det_input_dev(fh_det* d);          /* preprocessed input, NHWC with 4 lanes */
FH_API int fh_det_postprocess_dev(fh_det* d, int n, float score_thr, float nms_thr, fh_face* d_out, int max_per_frame,
                                  int* d_counts, void* stream);
/* FaceDetector::postprocess + nms (src/face_detector.cpp:224-338,356-384) on caller-supplied pre-decoded rows:
 * d_rows = [n][rows_per_frame][feat >= 15] fp32 (x1,y1,x2,y2,score,10 kps) in HBM; `scale` is the letterbox scale the
 * reference divides by.  Same kernels as the detector's own post-processing, without a graph in front: for callers that
 * decode elsewhere, and the hook through which the parity tests push crafted rows (ties, zero-area boxes, > 2048
 * survivors) through both branches of the NMS kernel.  Uses one process-wide scratch: calls must not overlap. */
FH_API int fh_postprocess_rows_dev(const float* d_rows, int n, int rows_per_frame, int feat, float scale, float score_thr,
                                   float nms_thr, fh_face* d_out, int max_per_frame, int* d_counts, void* stream);

/* ---- FaceRecognizer ----------------------------------------------------------------------
 * fh_rec_create          <- FaceRecognizer ctor + loadModel (src/face_recognizer.cpp:5-13,21-91)
 * fh_rec_extract         <- extractFeature(image, face)     (src/face_recognizer.cpp:236-304)
 * fh_rec_extract_simple  <- extractFeatureSimple(image)     (src/face_recognizer.cpp:152-234)
 *                           return the feature length written to out (L2-normalised), 0 for the
 *                           reference's empty-vector cases, < 0 on error.
 * fh_compare             <- compareFaces(f1, f2)            (src/face_recognizer.cpp:320-334)  */
FH_API fh_rec* fh_rec_create(const char* onnx_path);
FH_API void fh_rec_destroy(fh_rec* r);
FH_API int fh_rec_input_size(const fh_rec* r, int* width, int* height);
FH_API int fh_rec_feature_dim(const fh_rec* r);
FH_API double fh_rec_macs_per_face(const fh_rec* r);
FH_API double fh_rec_act_bytes_per_face(const fh_rec* r);
FH_API int fh_rec_set_chunk(fh_rec* r, int faces_per_pass);
FH_API int fh_rec_extract(fh_rec* r, const uint8_t* bgr, int rows, int cols, int step, const fh_face* face, float* out,
                          int out_cap);
FH_API int fh_rec_extract_simple(fh_rec* r, const uint8_t* bgr, int rows, int cols, int step, float* out, int out_cap);
FH_API float fh_compare(const float* f1, int n1, const float* f2, int n2);
/* n pre-aligned crops [n][H][W][3] BGR u8 in HBM -> d_out [n][dim] L2-normalised; d_raw (may be
 * NULL) receives the un-normalised network output. */
FH_API int fh_rec_embed_aligned_dev(fh_rec* r, const uint8_t* d_crops, int n, float* d_out, float* d_raw, void* stream);
/* Waits for `stream` and reports an error a launch of THIS handle raised after its asynchronous call had already returned (today: a
 * convolution hand-off that timed out, FH_ERR_DEVICE + fh_last_error()).  Such an error is otherwise returned by the next call
 * on the same handle; calls on other handles never see it.  FH_OK when the queued work completed. */
FH_API int fh_det_sync(fh_det* d, void* stream);
FH_API int fh_rec_sync(fh_rec* r, void* stream);
/* alignFace for n faces (d_frame_of[i] = frame index of face i, NULL = identity): writes crops
 * [n][H][W][3] and d_ok[n] (1 warped, 2 crop-resize fallback, 0 empty). */
FH_API int fh_rec_align_dev(fh_rec* r, const uint8_t* d_frames, int rows, int cols, int step, long long frame_stride,
                            const fh_face* d_faces, const int* d_frame_of, int n, uint8_t* d_crops, int* d_ok,
                            void* stream);
/* preprocessed network input of the last pass (NHWC, 4 lanes; only materialised with fh_rec_set_fused_stem(r, 0)) */
FH_API const float* fh_rec_input_dev(fh_rec* r);
FH_API int fh_rec_embed_faces_dev(fh_rec* r, const uint8_t* d_frames, int rows, int cols, int step,
                                  long long frame_stride, const fh_face* d_faces, const int* d_frame_of, int n,
                                  float* d_out, int* d_ok, void* stream);

/* ---- detect -> align -> embed on a batch of HBM-resident frames (the headline metric path).
 * Takes the first min(count, faces_per_frame) faces of each frame (score order) — the reference embeds "for every face"
 * (src/main.cpp:221-238).  d_faces / d_frame_of / d_emb must hold n*faces_per_frame entries; the compacted face list is
 * written front-to-back.  ONE host hand-off: after detect + NMS + selection the face count comes back through pinned memory
 * (the call waits for the detector only), and align + embed are then launched on exactly that many faces — dead slots cost
 * nothing.  Returns the number of faces (>= 0) or < 0; the embeddings are complete when `stream` has drained. */
FH_API int fh_pipeline_run_dev(fh_det* d, fh_rec* r, const uint8_t* d_frames, int n, int rows, int cols, int step,
                               long long frame_stride, float score_thr, float nms_thr, int faces_per_frame,
                               fh_face* d_faces, int* d_frame_of, float* d_emb, void* stream);

/* Two-stream form for streaming callers (the testWebcam loop shape, src/main.cpp:214-258, over batches): detect + decode +
 * NMS + face selection on stream_det, align + embed on stream_rec behind an event.  The host waits for the detector's face
 * count only; a recogniser queued earlier on stream_rec keeps running, so submitting batch k+1 straight after batch k
 * overlaps the HBM-bound detector with the MFMA-bound recogniser.  d_total (device int) also receives the count.  Every
 * buffer passed in (frames included) must stay untouched until stream_rec has drained; give each in-flight batch its own
 * d_faces / d_frame_of / d_emb / d_total.  Returns the number of faces or < 0. */
FH_API int fh_pipeline_submit_dev(fh_det* d, fh_rec* r, const uint8_t* d_frames, int n, int rows, int cols, int step,
                                  long long frame_stride, float score_thr, float nms_thr, int faces_per_frame,
                                  fh_face* d_faces, int* d_frame_of, float* d_emb, int* d_total, void* stream_det,
                                  void* stream_rec);

/* ---- streaming front end for HOST frames: the caller of the path (testWebcam, src/main.cpp:214-258: grab, detect, embed
 * every face) over batches.  The object owns a 2-slot ring of device buffers, a copy stream and a compute stream:
 * fh_stream_submit uploads the batch (tightly packed [n_frames][rows][cols][3] BGR u8; pinned memory makes the copy truly
 * asynchronous) while the previous batch is still computing, queues detect -> align -> embed behind it and returns the
 * batch's face count; fh_stream_collect waits for the OLDEST batch in flight and copies out its first min(count, cap)
 * faces / frame indices / embeddings (any of the three pointers may be NULL).  At most 2 batches in flight. */
typedef struct fh_stream fh_stream;
FH_API fh_stream* fh_stream_create(fh_det* d, fh_rec* r, int frames_per_batch, int rows, int cols, int faces_per_frame);
FH_API void fh_stream_destroy(fh_stream* s);
FH_API int fh_stream_submit(fh_stream* s, const uint8_t* host_frames, int n_frames, float score_thr, float nms_thr);
FH_API int fh_stream_collect(fh_stream* s, fh_face* faces, int* frame_of, float* emb, int cap);

/* ---- gallery (1:N compareFaces): rows are L2-normalised features; scores are (dot+1)/2. */
FH_API fh_gallery* fh_gallery_create(int dim);
FH_API void fh_gallery_destroy(fh_gallery* g);
FH_API int fh_gallery_upload(fh_gallery* g, const float* rows, long long n, int rows_on_device, long long index_base);
FH_API int fh_gallery_topk_dev(fh_gallery* g, const float* d_queries, int nq, int k, float* d_scores, int* d_indices,
                               void* stream);
/* Merge step of a row-SHARDED gallery (one shard per rank): d_part_scores / d_part_idx = [nparts][nq][k]
 * per-shard top-k lists (global row indices, -1 = empty slot) as one all-gather delivers them -> the overall top-k by
 * (score desc, index asc), the same total order and the same kernel fh_gallery_topk_dev finishes with, so a sharded
 * gallery returns exactly the single-gallery answer.  nparts * k <= 65536. */
FH_API int fh_topk_merge_dev(const float* d_part_scores, const int* d_part_idx, int nparts, int nq, int k, float* d_scores,
                             int* d_indices, void* stream);
/* ---- the exchange step of a row-sharded gallery behind the boundary (the reference's compareFaces loop,
 * src/face_recognizer.cpp:320-334 / src/main.cpp:221-238, over a gallery split across the GPUs of one node).  One process (or
 * thread) per rank; the collectives are RCCL (librccl, loaded on first use) on the CALLER's stream — no torch, no host copy.
 *   fh_comm_unique_id   rank 0 fills a 128-byte id; the caller hands it to the other ranks (any side channel: MPI, a file,
 *                       torch.distributed's store ...).
 *   fh_comm_create      binds this process to `device` (hipSetDevice) and joins the communicator; collective over all ranks.
 *                       Create it BEFORE the first fh_* object of the process where possible (RCCL sizes its buffers once).
 *   fh_gallery_topk_sharded_dev
 *                       every rank passes its nq_local query rows [nq_local][dim] (device) and its gallery shard `g` (uploaded
 *                       with its global index base): all-gather of the queries -> scan of the local shard for all
 *                       world * nq_local queries -> ONE all-gather of the per-rank (score, global index) lists -> merge
 *                       (the kernel fh_gallery_topk_dev finishes with).  d_scores / d_indices = [world * nq_local][k] on EVERY
 *                       rank, rank r's own queries in rows [r * nq_local, (r + 1) * nq_local): exactly the single-gallery
 *                       answer (score desc, index asc).  All ranks must call it with the same nq_local and k.  Asynchronous
 *                       on `stream`.  Returns world * nq_local, or < 0. */
#define FH_COMM_ID_BYTES 128
typedef struct fh_comm fh_comm;
FH_API int fh_comm_unique_id(unsigned char id[FH_COMM_ID_BYTES]);
FH_API fh_comm* fh_comm_create(int rank, int world, const unsigned char id[FH_COMM_ID_BYTES], int device);
FH_API void fh_comm_destroy(fh_comm* c);
FH_API int fh_comm_rank(const fh_comm* c);
FH_API int fh_comm_world(const fh_comm* c);
/* all-gather of equally sized float blocks on the caller's stream: d_recv = [world][count] (the frame-sharded callers use it for
 * per-rank embeddings; the sharded top-k uses it internally) */
FH_API int fh_comm_allgather_f32_dev(fh_comm* c, const float* d_send, float* d_recv, long long count, void* stream);
FH_API int fh_gallery_topk_sharded_dev(fh_gallery* g, fh_comm* c, const float* d_queries_local, int nq_local, int k,
                                       float* d_scores, int* d_indices, void* stream);
/* The webcam loop's reference handling (src/main.cpp:211-212,229-233,253-256) for an enrolled SET instead of one
 * refFeature: enroll appends rows (the 's' key; returns the index of the first new row), label gives every query its
 * best row when (dot+1)/2 > threshold ("Match", reference threshold 0.6, strict) and -1 otherwise ("Unknown");
 * d_scores[q] = the best mapped score (-1 for an empty gallery). */
FH_API long long fh_gallery_enroll(fh_gallery* g, const float* rows, long long n, int rows_on_device);
FH_API long long fh_gallery_size(fh_gallery* g);
FH_API int fh_gallery_label_dev(fh_gallery* g, const float* d_queries, int nq, float threshold, int* d_labels,
                                float* d_scores, void* stream);

/* ---- measurement hooks (bench.py): per-launch HIP-event timing of the network kernels.
 * Tags 0..3 = conv_igemm tile configs (128x128, 256x64, 128x32, 64x64), 4 = depthwise / depthwise+pointwise,
 * 5 = other graph ops, 6 = conv stream-K fix-up, 7 = Winograd GEMM (its FLOPs = executed; bytes slot = the layer's
 * direct-form FLOPs), 8 = Winograd transforms, 9 = spatial-tile (LDS halo) 3x3 convolutions, 10 = conv_tall_kernel (whole tile rounds of the 3x3 stride-1 layers), 11 = conv_pw_kernel (1x1 stride-1 convolutions as plain GEMMs), 12 = wino2_kernel (fused Winograd F(2x2,3x3): FLOPs = executed, bytes slot = the
 * layer's direct-form FLOPs).  fh_timing_num_tags() = 13 today; fh_timing_collect synchronises, fills n >= fh_timing_num_tags() entry arrays (elapsed ms,
 * algorithmic FLOP, algorithmic activation bytes, launches) and resets the counters.
 * fh_*_set_conv_cfg forces one tile config for every dense conv of a handle (-1 = automatic)
 * and switches the stream-K remainder wave on/off (tuning / A-B measurements). */
FH_API int fh_timing_enable(int on);
FH_API int fh_timing_num_tags(void);
FH_API int fh_timing_collect(double* ms, double* flops, double* bytes, long long* launches, int n);
FH_API int fh_timing_collect_ops(double* ms, double* flops, int* tag, int cap);   /* per launch, in order */
FH_API int fh_det_set_conv_cfg(fh_det* d, int cfg, int stream_k);
FH_API int fh_rec_set_conv_cfg(fh_rec* r, int cfg, int stream_k);
/* 3x3 stride-1 convolutions with >= 128 input channels run as Winograd F(4x4,3x3) (a quarter of the matrix-core work,
 * fp32 rounding error ~25x the direct form's: see DESIGN.md) unless switched off here; 0 = direct form everywhere. */
FH_API int fh_det_set_winograd(fh_det* d, int on);
FH_API int fh_rec_set_winograd(fh_rec* r, int on);
/* on (default): between two consecutive Winograd layers on a map of <= 16x16 pixels the output transform of the first and the input
 * transform of the second run as one kernel (the activation stays in LDS); off: separate transform kernels. */
FH_API int fh_rec_set_wino_fusion(fh_rec* r, int on);
/* on (default): the strided 1x1 shortcut convolution of an IResNet block runs as a tenth tap inside the K loop of the 3x3 convolution
 * it is added to (weights concatenated along K, one launch, no residual round trip); off: two convolutions + residual add. */
FH_API int fh_rec_set_shortcut_fold(fh_rec* r, int on);
/* Opt-in precision mode of the recogniser (the default and the headline stay fp32 = the reference's own arithmetic,
 * src/face_recognizer.cpp:58 float tensors through onnxruntime).  FH_PREC_BF16X2: the Winograd GEMMs (the 3x3 convolutions with >= 128
 * channels, ~80% of IResNet-50's FLOPs) take each operand as a (hi, mid) pair of bf16 — 16 mantissa bits — and run three bf16 MFMAs
 * with f32 accumulation in place of one f32 MFMA; transforms, epilogues, all other layers and the embedding stay fp32.
 * The call is GATED: it embeds a fixed pseudo-random batch of 64 crops in both modes and enters the mode only if every pair of
 * embeddings agrees to 1 - cos < 1e-3; otherwise it returns FH_ERR_STATE, the handle stays fp32 and fh_last_error() quotes the measured
 * value.  *worst (may be NULL) receives the measured max(1 - cos).  Returns the number of layers switched (>= 1), or < 0. */
enum fh_precision { FH_PREC_FP32 = 0, FH_PREC_BF16X2 = 1 };
FH_API int fh_rec_set_precision(fh_rec* r, int mode, float* worst);
FH_API int fh_rec_get_precision(fh_rec* r);
/* A handle whose stream is restricted to a subset of the CUs (hipExtStreamCreateWithCUMask, e.g. detector and
 * recogniser side by side on disjoint CU sets) should say how many it gets: it sizes the convolution kernels'
 * remainder round.  0 = the whole device (default). */
FH_API int fh_det_set_cus(fh_det* d, int cus);
FH_API int fh_rec_set_cus(fh_rec* r, int cus);
/* on (default): the u8 preprocess is fused into the first convolution and the preprocessed input tensor is
 * never materialised; off: separate preprocess kernel (needed for fh_det_input_dev). */
FH_API int fh_det_set_fused_stem(fh_det* d, int on);
FH_API int fh_rec_set_fused_stem(fh_rec* r, int on);
/* on (default): when the graph opens with conv 3x3 (16 channels) -> depthwise 3x3 -> pointwise 1x1 (SCRFD's first block), the stem
 * is computed inside the depthwise -> pointwise kernel and its output map never reaches memory; off: separate kernels. */
FH_API int fh_det_set_fused_front(fh_det* d, int on);
/* on (default): dense 3x3 stride-1 convolutions with 16 input and <= 64 output channels on large maps (SCRFD's FPN and head
 * convolutions) run on 8x16 spatial tiles with an LDS halo (conv_halo.hip); off: the generic implicit-GEMM kernel. */
FH_API int fh_det_set_halo_conv(fh_det* d, int on);

/* ---- image files -> BGR u8, replaces cv::imread(path) (reference src/main.cpp:42,71-72,140-141; OpenCV's default
 * IMREAD_COLOR: 8-bit BGR, alpha dropped, grey replicated, JPEG EXIF orientation applied).  Host code.  JPEG
 * (baseline + progressive; libjpeg's islow IDCT / fancy up-sampling / YCbCr tables restated), PNG, BMP, PPM/PGM.
 * *bgr is malloc'ed [rows*cols*3], release it with fh_image_free.  0 on success, < 0 + fh_last_error(). */
FH_API int fh_imread(const char* path, unsigned char** bgr, int* rows, int* cols);
FH_API int fh_image_decode(const unsigned char* bytes, size_t n, unsigned char** bgr, int* rows, int* cols);
FH_API void fh_image_free(unsigned char* bgr);

/* ---- single kernels exposed for parity tests and micro-benchmarks (device pointers). */
FH_API int fh_memcpy_d2h(void* host_dst, const void* dev_src, size_t bytes);   /* synchronous */
FH_API int fh_resize_u8c3_dev(const uint8_t* d_src, int sh, int sw, int sstep, uint8_t* d_dst, int dh, int dw, int dstep,
                              void* stream);
FH_API int fh_conv_forward_dev(const float* d_in, const float* d_wt_packed, const float* d_bias, float* d_out, int batch,
                               int h, int w, int cin, int cout, int ksize, int stride, int kpad, int cfg, void* stream);
/* 3x3 stride-1 pad-1 convolution (+bias) in the Winograd F(4x4,3x3) form the deep layers use; w_ohwi = HOST weights
 * [cout][3*3][cin]; cin % 32 == 0, cout % 4 == 0.  Synchronous. */
FH_API int fh_conv_winograd_dev(const float* d_in, const float* w_ohwi_host, const float* d_bias, float* d_out, int batch, int h, int w,
                                int cin, int cout, void* stream);
/* The same convolution in the fused Winograd F(2x2,3x3) form the 64-channel stages use (conv_wino2.hip): cin == 64, cout % 64 == 0;
 * d_bias = [cout] or, with bias_cls != 0, [9][cout] (one vector per border class of the output pixel: a pre-conv BatchNorm folded in);
 * act = 0 none / 1 ReLU / 2 PReLU (d_slope [cout]) ...; d_res = optional residual of the output's shape.  Synchronous. */
FH_API int fh_conv_wino2_dev(const float* d_in, const float* w_ohwi_host, const float* d_bias, const float* d_slope, const float* d_res,
                             float* d_out, int batch, int h, int w, int cin, int cout, int act, int bias_cls, void* stream);
/* The full argument set of that kernel: d_out2 = d_out * d_s2 + d_t2 (a following block's BatchNorm; any of d_out / d_out2 may be NULL
 * when the other is given), and the MERGED form SCRFD's heads use — n_outs (1..3) sibling convolutions evaluated as one with cout <= 32
 * channels in all: output g takes channels [oc0[g], oc0[g + 1]) into d_outs[g] ([pixels][oc0[g + 1] - oc0[g]]) through activation
 * oact[g] (0 none / 1 ReLU / 3 sigmoid); no residual, second output or bias classes in that form.  Synchronous. */
FH_API int fh_conv_wino2_ex_dev(const float* d_in, const float* w_ohwi_host, const float* d_bias, const float* d_slope, const float* d_res,
                                float* d_out, float* d_out2, const float* d_s2, const float* d_t2, int n_outs, float* const* d_outs,
                                const int* oc0, const int* oact, int batch, int h, int w, int cin, int cout, int act, int bias_cls,
                                void* stream);
/* Diagnostic builds of conv_wino2.hip (-DFACEHIP_W2_PROF, scripts/wino2_prof.sh, FACEHIP_W2_ABLATE set): median shader clock in MHz the
 * waves of the last wino2 launch measured (s_memtime against the 100 MHz s_memrealtime).  0 in production builds. */
FH_API double fh_debug_wino2_clock_mhz(void);
/* Batch-1 host-pointer calls (fh_det_detect, fh_rec_extract, fh_rec_extract_simple — the reference's own mode, src/main.cpp:88-104) are
 * captured into a HIP graph per call shape (image size / pitch, thresholds) and replayed: first call with a shape eager, second
 * captured, later ones one hipGraphLaunch each.  Results are bitwise those of the eager path.  fh_set_graph_replay(0) (or
 * FACEHIP_GRAPH=0) turns it off process-wide; fh_*_graph_stats return the node count of the handle's captured graph (0 = none yet)
 * and the number of replayed calls. */
FH_API int fh_set_graph_replay(int on);
FH_API const float* fh_det_workspace_dev(fh_det* d);          /* tuning hook: the detector's stream-K workspace (diagnostic kernel builds write phase stamps there) */
FH_API int fh_det_graph_stats(fh_det* d, long long* replays);
FH_API int fh_rec_graph_stats(fh_rec* r, long long* replays);
/* Stream-K watchdog test hook (conv_mfma.hip): drop_publish != 0 makes the helper workgroups of a remainder round "lose" their
 * publication, timeout_ms bounds the owners' wait (0 = the 2 s default).  An owner whose wait times out abandons its tile and the
 * next call on the same handle (or fh_det_sync / fh_rec_sync) returns FH_ERR_DEVICE ("stream-K hand-off timed out ...") instead of
 * the process hanging with the GPU; other handles are unaffected. */
FH_API int fh_debug_streamk(int drop_publish, int timeout_ms);
/* Test hook of the multi-tile Winograd GEMM (winograd.hip wino_gemm_pers_kernel): launches with more tiles than resident workgroup
 * slots walk several tiles per workgroup; slots > 0 makes the launcher pretend the device has that many (rounded up to 8), so that small
 * test layers take the multi-tile path with many tiles per workgroup; 0 restores the device's own count. */
FH_API int fh_debug_wino_slots(int slots);
FH_API int fh_conv_wt_rows(int cout);
/* host: weights [cout][ksize*ksize][cin] (O,H,W,I) -> the kernel's packed image [fh_conv_wt_rows][fh_conv_kpad] */
FH_API int fh_conv_pack_weights(const float* w_ohwi, int cout, int cin, int ksize, float* dst_packed);
FH_API int fh_conv_kpad(int ktot);

#ifdef __cplusplus
}
#endif
#endif /* FACEHIP_H_ */
