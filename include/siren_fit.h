/*
 * siren_fit.h — C ABI of libsiren_fit.so, the MI355X (gfx950) SIREN fitting engine.
 *
 * One handle == one per-image fit == one HIP stream on one device.  The library replaces
 * the arithmetic of ONE hot path of varun19299/implicit-image-compression (citations are
 * relative to the reference tree):
 *
 *   implicit_image/data.py:78-88              get_grid         -> sf_set_coords (two 1-D linspace vectors)
 *   implicit_image/models/siren.py:56-68      SineLayer.forward \
 *   implicit_image/models/siren.py:123-134    Siren.forward      > sf_forward / sf_forward_backward
 *   implicit_image/utils/train_helper.py:147-161  mse_loss + backward (autograd) /
 *   implicit_image/utils/train_helper.py:166-177  optimiser step (torch.optim.Adam, conf/optim/adam.yaml)
 *   implicit_image/pipeline/masking/core.py:271-279,671-702  Masking.step -> apply_mask   -> sf_adam_step
 *   implicit_image/utils/train_helper.py:41-59    eval_epoch (fwd, MSE)                   -> sf_forward
 *
 * Conventions
 *   - every function returns 0 on success, a negative sf_status otherwise; the message is
 *     available from sf_last_error() (never throws, never exits);
 *   - pointers named *_dev are DEVICE pointers owned by the caller (e.g. torch tensor.data_ptr());
 *     the engine never frees them.  sf_set_target BORROWS its pointer: the image must stay
 *     alive and unchanged until the next sf_set_target or sf_destroy;
 *   - "flat" parameter order is the reference's named_parameters() order
 *     (layers.0.linear.weight, layers.0.linear.bias, layers.1.linear.weight, ...), weights in
 *     nn.Linear layout [out][in] row-major, fp32;
 *   - all work is enqueued on the handle's stream; functions that return host scalars
 *     (loss_out / sse_out != NULL) synchronise that stream, the others do not;
 *   - a handle is not thread-safe; distinct handles are independent.
 */
#ifndef SIREN_FIT_H_
#define SIREN_FIT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SF_ABI_VERSION 2

typedef enum sf_status {
  SF_OK = 0,
  SF_ERR_INVALID = -1,     /* bad argument / unsupported configuration */
  SF_ERR_HIP = -2,         /* a HIP runtime call failed                */
  SF_ERR_NO_DEVICE = -3,   /* no gfx950 device visible                 */
  SF_ERR_STATE = -4,       /* call order violated (e.g. step before set_target) */
  SF_ERR_NOMEM = -5
} sf_status;

/* arithmetic type of the hidden-layer GEMM operands (accumulation is always fp32,
 * layer 0 and all optimiser state are always fp32) */
typedef enum sf_dtype {
  SF_BF16 = 0,   /* bf16 operands, 1 MFMA per product                                  */
  SF_F16 = 1,    /* fp16 operands (11-bit significand), gradients pre-scaled            */
  SF_BF16X3 = 2  /* forward operands split hi+lo in bf16 (3 MFMAs per forward product)  */
} sf_dtype;

typedef struct sf_config {
  int32_t abi_version;      /* SF_ABI_VERSION                                            */
  int32_t height, width;    /* full image H, W: the loss mean is over 3*H*W (train_helper.py:151) */
  int32_t row_begin, row_end; /* rows [row_begin,row_end) fitted by THIS handle (pixel-split); 0,H = all */
  int32_t in_features;      /* 2  (siren.py:74)  */
  int32_t out_features;     /* 3  (siren.py:75)  */
  int32_t hidden;           /* mlp.hidden_size after the small-dense scaling (siren.py:88) */
  int32_t depth;            /* number of Linear layers incl. first and last (siren.py:90-118) */
  float first_omega_0;      /* conf/mlp/siren.yaml:5 */
  float hidden_omega_0;     /* conf/mlp/siren.yaml:6 */
  int32_t outermost_linear; /* 1: last layer has no sine (siren.py:114) */
  int32_t compute_dtype;    /* sf_dtype */
  float beta1, beta2, eps;  /* Adam (torch.optim.Adam defaults 0.9, 0.999, 1e-8) */
  int32_t device;           /* HIP device ordinal */
  void* stream;             /* hipStream_t to enqueue on (NULL = the device's null stream) */
  int64_t chunk_pixels;     /* pixels processed per kernel sweep (0 = auto) */
  int32_t scratch_format;   /* width of the two tensors the backward re-reads from HBM (phases, deltas):
                             * 16 = unorm16 phases + 16-bit float deltas (round-1 format);
                             * 12 = phase BYTES + 16-bit float deltas;
                             *  8 = phase bytes + fp8 e4m3 deltas under one power-of-two scale per pixel chunk (from the
                             *      chunk's residual) times one per layer (from the layer's weight norm);
                             *  0 = auto, SF_F16 only (8 and 12 need fp16 operands; else 16): hidden <= 256: 8 for images of
                             *      >= 2^20 pixels, 12 below; hidden > 256: from 2^20 pixels 8 up to hidden 512 and 12 above
                             *      (8 is accepted there when asked for), 16 below.  An auto handle moves to 16 when sf_set_masks sets a mask
                             *      (sparse networks: DESIGN.md section 2)                                             */
} sf_config;

typedef struct sf_engine sf_handle;

/* lifecycle */
int sf_create(const sf_config* cfg, sf_handle** out);
int sf_destroy(sf_handle* h);
const char* sf_last_error(void);            /* thread-local message of the last failure */
int sf_abi_version(void);

/* shapes */
int sf_num_params(const sf_handle* h, int64_t* n_params);          /* P, length of every flat vector */
int sf_scratch_format(const sf_handle* h, int32_t* format);        /* the format in use (8 / 12 / 16): what 0 resolved to */
int sf_param_offset(const sf_handle* h, int32_t layer, int64_t* weight_off, int64_t* bias_off);

/* model state: flat fp32 vectors of length P on the device */
int sf_set_params(sf_handle* h, const float* flat_dev);
int sf_get_params(sf_handle* h, float* flat_dev);
int sf_set_masks(sf_handle* h, const float* flat_dev);   /* 0/1 per parameter (1 for biases); NULL = dense */
int sf_get_grads(sf_handle* h, float* flat_dev);         /* dense gradient of the last forward_backward   */
int sf_set_grads(sf_handle* h, const float* flat_dev);   /* e.g. after an all-reduce over ranks           */
int sf_get_adam_state(sf_handle* h, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t* step);
int sf_set_adam_state(sf_handle* h, const float* exp_avg_dev, const float* exp_avg_sq_dev, int64_t step);
/* direct device views of engine-owned state (valid until sf_destroy); which: 0 params, 1 grads,
 * 2 exp_avg, 3 exp_avg_sq, 4 masks */
int sf_state_ptr(sf_handle* h, int32_t which, float** dev_ptr);
/* device address of the engine's sum-of-squared-residuals scalar (double) that the last sf_forward /
 * sf_forward_backward wrote on the handle's stream: pixel-split ranks all-reduce it together with the gradient
 * view instead of synchronising for the host value */
int sf_sse_ptr(sf_handle* h, double** dev_ptr);
/* test / debugging aid: device address and size in bytes of an engine scratch tensor of the last pass.
 * which: 0 phases (all layers, layer stride = bytes / (depth-1)), 1 deltas, 2 dL/dout, 3 per-workgroup slabs */
int sf_debug_scratch(sf_handle* h, int32_t which, void** dev_ptr, int64_t* bytes);
/* 1-D k-means quantisation of one weight tensor on the handle's stream, NO host synchronisation.  Replaces
 * pipeline/quant/kmeans.py:110-150 (find_centroids) + kmeans_helper.py:59-115 (kmeans_fit / kmeans_predict).
 *   w_dev [n]            the weights (device); zeros are excluded from the fit, included in the prediction
 *   centers_dev [K]      in: the initial guess - torch.linspace(min, max, K) over the non-zero weights, K = 2^bits - 1
 *                        (kmeans.py:125-131); out: the Lloyd centres after <= iter_limit iterations (tol on (sum |dc|)^2)
 *   centroids_dev [cap]  out: {0} U centres -> unique -> ordered by |c|, zero padded (cap >= K + 1)
 *   n_centroids_dev      out (device int, may be NULL): how many of them are real
 *   labels_dev [n]       out (may be NULL): argmin of the squared distance, first index wins (int64, as torch.argmin)
 *   new_weight_dev [n]   out (may be NULL): centroids[labels] */
int sf_kmeans_fit(sf_handle* h, const float* w_dev, int64_t n, float* centers_dev, int32_t K, int32_t iter_limit, float tol,
                  float* centroids_dev, int32_t centroids_cap, int32_t* n_centroids_dev, int64_t* labels_dev,
                  float* new_weight_dev);
/* test aid for the "never throws" promise above: raises inside the library on purpose (0: std::bad_alloc -> SF_ERR_NOMEM,
 * 1: std::runtime_error, 2: a non-std exception -> SF_ERR_INVALID); every entry point is a function-try-block */
int sf_debug_throw(int32_t kind);
/* tell the engine that the caller wrote the parameters through the sf_state_ptr(…,0) view */
int sf_params_changed(sf_handle* h);

/* data: the two linspace vectors of get_grid (data.py:82-83) and the target image rows.  rows / cols must be
 * torch.linspace(0, 1, n) to within 2e-6 (the layer-0 gradient kernels re-derive coordinates as i/(n-1));
 * anything else returns SF_ERR_INVALID */
int sf_set_coords(sf_handle* h, const float* rows_dev /*[height]*/, const float* cols_dev /*[width]*/);
int sf_set_target(sf_handle* h, const float* img_dev /*[(row_end-row_begin)*width*3], borrowed*/);

/* the hot path */
/* forward only (eval_epoch): pred_dev may be NULL; sse_out (host) = sum of squared residuals over
 * this handle's rows (double), NULL = do not synchronise */
int sf_forward(sf_handle* h, float* pred_dev, double* sse_out);
/* forward + loss + backward: leaves the dense gradient (already scaled by 1/(3*H*W)) in the
 * engine; sse_out as above */
int sf_forward_backward(sf_handle* h, double* sse_out);
/* Adam (+ mask) on the current gradient with learning rate lr; refreshes the low-precision weight images */
int sf_adam_step(sf_handle* h, float lr);
/* n_steps x (forward_backward + adam_step) with learning rates lr[0..n_steps) (host array);
 * loss_out (host, may be NULL) receives the MSE of every step (length n_steps) */
int sf_step(sf_handle* h, const float* lr, int32_t n_steps, float* loss_out);
/* sf_step execution mode for single-chunk fits: 0 (default) = one stream launch per kernel, 1 = capture one
 * training step into a hipGraph and replay it n_steps times.  Results are bit-identical; which is faster is a
 * property of the runtime (measured on ROCm 7.2 / MI355X: eager 73 us/step vs replay 87 us/step at SIREN 64x4
 * on 256x256), so replay stays opt-in. */
int sf_set_graph_replay(sf_handle* h, int32_t on);

/* measurement: per-kernel HIP-event timing on the handle's stream */
int sf_profile_enable(sf_handle* h, int32_t on);
int sf_profile_reset(sf_handle* h);
int sf_profile_num_kernels(const sf_handle* h, int32_t* n);
int sf_profile_get(sf_handle* h, int32_t idx, const char** name, double* total_ms, int64_t* launches,
                   double* flops_per_launch, double* bytes_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* SIREN_FIT_H_ */
