/* wn_abi.h -- C ABI of libwn_mi355.so, the MI355X (gfx950) fast-generation engine.
 *
 * This is the drop-in boundary for ONE hot path of vincentherrmann/pytorch-wavenet:
 * WaveNetModel.generate_fast() and everything it calls per timestep.  The reference has no FFI seam
 * (it is pure Python); each entry point below names the reference interface it replaces
 * (paths relative to the reference root).  The Python facade (pytorch-wavenet_amd/wavenet_model.py)
 * binds these with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 (WN_OK) or a negative WN_E_* code and never
 *     throws; wn_last_error() returns a thread-local human-readable message for the last failure.
 *   - "device pointer" = HIP device memory on the handle's device (e.g. torch.Tensor.data_ptr()),
 *     "host pointer"   = ordinary process memory.
 *   - a handle is single-threaded (like a reference model object whose DilatedQueues are mutable
 *     state); distinct handles may be used concurrently from different threads (the reference calls
 *     generate_fast from a daemon thread during training: model_logging.py:48-58).
 *   - wn_generate jobs are persistent kernels: every workgroup of a job must be resident before the job
 *     makes progress, and two jobs that do not fit the chip together would each get part of it and
 *     neither would ever run.  Two mechanisms make residency a requirement instead of a hope:
 *     (1) ADMISSION.  wn_generate books the job's CUs (per XCD: the workgroups it keeps resident on its
 *     fullest XCD) in a per-device table -- a file named after the device's PCI bus id, used under
 *     flock(): /dev/shm/wn_mi355_gate_u<euid>_<busid>, mode 0600, shared by the processes of ONE user;
 *     WN_GATE_DIR=<dir> names a directory an administrator prepared for all users of the device
 *     (<dir>/wn_mi355_gate_<busid>).  The file is opened O_NOFOLLOW, created O_EXCL, and trusted only
 *     after fstat() (regular, one link, the user's own, not world-writable); a table that fails the checks
 *     is reported once on stderr and the device's gate is process-local from then on (wn_info.gate_shared
 *     = 0).  A job whose booking does not fit WAITS -- first come first served, bounded by
 *     WN_GATE_TIMEOUT_MS, default 10 minutes, then WN_E_TIMEOUT -- until the jobs in front of it have
 *     finished: two cfg3 jobs (28 of 32 CUs per XCD each) from two threads or two processes run one after
 *     the other, two cfg1 jobs side by side.  Jobs one HIP stream already serialises (same process, same
 *     stream) share a booking: wn_generate stays asynchronous for them.  The booking is returned when the
 *     kernel finishes (a host function enqueued behind it), at the latest in wn_wait; bookings of
 *     processes that no longer exist are dropped.
 *     (2) RESIDENCY BARRIER.  wn_create checks the plan against what the device can hold of the job's
 *     kernel (hipOccupancyMaxActiveBlocksPerMultiprocessor), wn_generate against the stream's CU mask;
 *     and every workgroup of a job checks in at start-up and enters the chain only when ALL have (kernels
 *     that are not wn_generate jobs -- a long torch kernel -- are not booked: a job that finds CUs taken
 *     by one starts when they free up).  That wait has its own bound (WN_RESIDENT_TIMEOUT_MS, default 60 s)
 *     and its own error, WN_E_BUSY: nothing ran, queues unchanged, the call can be repeated.  The per-hand-off
 *     bound (timeout_ms -> WN_E_TIMEOUT) only starts behind the barrier.
 *     A job is ONE persistent kernel (any stream count up to ~150 at cfg3's shape), or, beyond one chain's
 *     capacity, rounds of up to 128 streams, one kernel after the other on the caller's stream.
 */
#ifndef WN_ABI_H
#define WN_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WN_ABI_VERSION 5  /* 5: wn_train_pack / wn_train_unpack_grads (wn_train_tensors), wn_train_set_deterministic, wn_adam_args.flags (several parameter groups
                                clipped together); a job of several rounds reports WN_E_BUSY only when NO round started (mixed outcomes: WN_E_STATE until wn_reset);
                             4: WN_E_BUSY (start-up residency barrier with its own bound, WN_RESIDENT_TIMEOUT_MS), wn_info.forward_native, wn_adam_step;
                             3: per-device admission of persistent jobs (wn_info: gate_*), kernel variant 4 (layers_per_workgroup);
                             2: wn_train_loss; wn_info reports the form of the chain (streams_per_item, head_replicas, n_samplers) */

enum {
    WN_OK = 0,
    WN_E_BADARG = -1,      /* NULL / out-of-range argument */
    WN_E_UNSUPPORTED = -2, /* configuration does not fit this device (see wn_last_error) */
    WN_E_HIP = -3,         /* a HIP runtime call failed (message carries hipGetErrorString) */
    WN_E_NOMEM = -4,
    WN_E_TIMEOUT = -5,     /* the persistent kernel gave up waiting on a hand-off (bounded spins) */
    WN_E_STATE = -6,       /* call order violated, e.g. generate before load_weights */
    WN_E_BUSY = -7         /* the job's workgroups did not all become resident (CUs held by other kernels): nothing ran, the call can be repeated */
};

/* Mirrors the constructor of WaveNetModel (wavenet_model.py:28-39) plus engine placement. */
typedef struct wn_config {
    int32_t layers;            /* layers per block                    wavenet_model.py:29 */
    int32_t blocks;            /*                                      :30 */
    int32_t dilation_channels; /* D                                    :31 */
    int32_t residual_channels; /* R                                    :32 */
    int32_t skip_channels;     /* S                                    :33 */
    int32_t end_channels;      /* E                                    :34 */
    int32_t classes;           /* C (mu-law classes, 256)              :35 */
    int32_t kernel_size;       /* k taps of the dilated convs          :37 */
    int32_t bias;              /* stack convs carry a bias             :39 */
    int32_t n_streams;         /* independent generation streams; the reference API has exactly 1 */
    int32_t device_id;         /* HIP device ordinal */
    int32_t layer_split;       /* workgroups (CUs) that share one layer; 0 = choose automatically */
    int32_t head_split;        /* workgroups that share end_conv_1/end_conv_2; 0 = automatic */
    int32_t reserved[3];       /* [0]: bit 0 = WN_CFG_NO_PADDING, every other bit 0; [1], [2]: must be 0 */
} wn_config;

/* Host pointers to fp32 parameters in the reference's nn.Conv1d layout (out, in, k), the per-layer
 * tensors concatenated along a leading NL = layers*blocks axis (state_dict names in comments,
 * wavenet_model.py:59-119).  Bias pointers of the stack may be NULL when cfg.bias == 0. */
typedef struct wn_weight_ptrs {
    const float* start_w;  /* start_conv.weight        (R, C, 1)      */
    const float* start_b;  /* start_conv.bias          (R)      | NULL */
    const float* filter_w; /* filter_convs.N.weight    (NL, D, R, k)  */
    const float* filter_b; /* filter_convs.N.bias      (NL, D)  | NULL */
    const float* gate_w;   /* gate_convs.N.weight      (NL, D, R, k)  */
    const float* gate_b;   /* gate_convs.N.bias        (NL, D)  | NULL */
    const float* res_w;    /* residual_convs.N.weight  (NL, R, D, 1)  */
    const float* res_b;    /* residual_convs.N.bias    (NL, R)  | NULL */
    const float* skip_w;   /* skip_convs.N.weight      (NL, S, D, 1)  */
    const float* skip_b;   /* skip_convs.N.bias        (NL, S)  | NULL */
    const float* end1_w;   /* end_conv_1.weight        (E, S, 1)      */
    const float* end1_b;   /* end_conv_1.bias          (E)            */
    const float* end2_w;   /* end_conv_2.weight        (C, E, 1)      */
    const float* end2_b;   /* end_conv_2.bias          (C)            */
} wn_weight_ptrs;

/* One generate_fast()-shaped job: n_given-1 priming evaluations followed by num_samples generating
 * evaluations, for every stream (wavenet_model.py:259-311).  All array pointers are DEVICE pointers. */
typedef struct wn_generate_args {
    const int32_t* first_samples; /* [n_streams][n_given] given class indices (first_samples, :245-257)   */
    int64_t n_given;              /* >= 1; the last given sample is the input of generating step 0        */
    int64_t num_samples;          /* >= 0                                                                 */
    float temperature;            /* > 0: sample from softmax(x/T) (:282-289);  <= 0: argmax (:290-294)   */
    int32_t flags;                /* 0 */
    const float* regularizer;     /* [classes] fp32 values subtracted from the logits (:273-274,280) | NULL */
    const double* uniforms;       /* [n_streams][num_samples] U[0,1) draws, one per generated sample, the
                                     host's np.random.random_sample() stream (np.random.choice, :288);
                                     NULL => greedy regardless of temperature                              */
    int32_t* out_idx;             /* [n_streams][num_samples] generated class indices                      */
    float* dbg_logits;            /* [n_streams][num_samples][classes] pre-regulariser logits | NULL       */
    void* hip_stream;             /* hipStream_t to enqueue on (NULL = default stream)                     */
    int32_t timeout_ms;           /* per-hand-off spin bound inside the kernel; 0 = default (10 s)         */
    int32_t reserved;
    const float* stream_temperatures; /* [n_streams] fp32 | NULL: per-stream temperature overriding `temperature`
                                     (<= 0: that stream takes the argmax) -- generate_audio's list of
                                     temperatures (wavenet_training.py:115-124) as parallel streams          */
} wn_generate_args;

/* What the planner decided (for logs, benches and tests). */
typedef struct wn_info {
    int32_t abi_version;
    int32_t n_layers;        /* NL */
    int32_t layer_split;     /* P  */
    int32_t head_split;      /* PA */
    int32_t n_workgroups;    /* NL*P + PA persistent workgroups = CUs used */
    int32_t lds_bytes;       /* dynamic LDS per workgroup */
    int32_t n_compute_units; /* CUs on the device */
    int32_t receptive_field; /* wavenet_model.py:53,106-107 */
    int64_t weight_bytes;    /* packed weight banks resident in HBM (copied to LDS at launch) */
    int64_t queue_bytes;     /* dilation-queue rings */
    int64_t handoff_bytes;   /* inter-workgroup granule buffers */
    int64_t evals_done;      /* timesteps evaluated since the last wn_reset (queue time) */
    int32_t kernel_variant;  /* 1 = generic kernel (weights stationary in LDS, any shape)
                                3 = wave-specialised kernel (768-thread layer workgroups: critical / skip / queue wave groups, weights
                                    stationary in registers, one chain for up to ~150 streams; from n_layers + 6 streams two streams per
                                    layer item and two replicas of the head workgroups): every instantiated shape -- BASELINE configs 1-4
                                    and the train_script.py shape
                                4 = stacked-layer kernel (2-5 consecutive layers per 512-thread workgroup, the layer-to-layer hand-off in LDS;
                                    layers_per_workgroup says how many): the small shapes -- cfg1, cfg2, the train_script.py shape -- at few
                                    streams (up to (stack workgroups + 2) / 2)
                                (2 = the 256-thread register-resident kernels of ABI version 1: removed) */
    int32_t n_chains;        /* 1, or the rounds of up to 128 streams a job beyond one chain's capacity runs one after the other;
                                n_workgroups and the byte counts are totals over them */
    int32_t streams_per_item; /* the FORM that runs (variant 3; 1 elsewhere): streams a layer workgroup processes per pipeline item */
    int32_t head_replicas;    /* ... replicas of the head workgroups (replica j serves the streams s = j mod head_replicas) */
    int32_t n_samplers;       /* ... dedicated sampler workgroups (0: layer 0 samples itself, single-stream kernels of variant 1 / 2) */
    int32_t dev_overrides;    /* 1 iff WN_TESTING=1 let a development override (WN_KERNEL, WN_V3_MODE, ...) change what the planner chose */
    int32_t layers_per_workgroup; /* variant 4: consecutive layers one stack workgroup holds (hand-offs between them stay in LDS); 1 elsewhere */
    int32_t gate_shared;      /* admission table of the LAST job: 1 = the shared table (see the header comment), 0 = this process only
                                 (no usable table: said once on stderr), -1 = no job yet */
    int32_t gate_waited_ms;   /* how long the last job waited for jobs booked in front of it */
    int32_t gate_need_per_xcd; /* CUs per XCD a job of this handle books: the workgroups it keeps resident on its fullest XCD (of n_compute_units / 8) */
    int32_t forward_native;   /* after wn_load_weights: 1 = wn_forward / wn_prime serve this handle's (padded) shape, 0 = they answer WN_E_UNSUPPORTED
                                 for every call (kernel_size != 2, channel counts that are not multiples of 32 after padding); the facade keys its
                                 "do not ask again" on THIS, not on the text of an error */
    int32_t workgroups_per_cu; /* workgroups of the job's kernel ONE compute unit holds with the job's LDS (hipOccupancyMaxActiveBlocksPerMultiprocessor,
                                  checked against the plan at wn_create) */
    int32_t resident_timeout_ms; /* bound of the start-up residency barrier of the LAST job (WN_RESIDENT_TIMEOUT_MS, default 60 s); 0 = no job yet */
    int32_t skip_lane_slots;   /* variant 3: hand-off slots of the skip lanes re-used per in-flight item (the throughput-bound form of cfg3's kernel, 96 streams
                                  and more); 0 = one slot per stream */
} wn_info;

/* wn_config.reserved[0]: plan the model's OWN channel shape (no zero padding into a compiled kernel shape, see wn_create): what a
 * handle that serves wn_train_* needs, whose parameter layout must be the caller's. */
#define WN_CFG_NO_PADDING 1

typedef struct wn_handle wn_handle;

int wn_abi_version(void);

/* WaveNetModel.__init__ (wavenet_model.py:28-123): plans the workgroup chain, allocates queues/hand-off
 * buffers on cfg->device_id.  No weights yet.
 * Channel counts the wave-specialised kernel is not compiled for are served by zero padding (kernel_size 2, 256 classes,
 * no pinned split): the handle then runs the next instantiated shape that holds the model, wn_load_weights pads the
 * caller's arrays with zeros, results are those of the caller's model, wn_export_queue returns the caller's channels,
 * wn_train_* return WN_E_UNSUPPORTED (weight_bytes / queue_bytes of wn_info are the padded model's).
 * cfg->reserved[0] & WN_CFG_NO_PADDING switches the padding off. */
int wn_create(const wn_config* cfg, wn_handle** out);
void wn_destroy(wn_handle* h);

/* nn.Module.load_state_dict / the parameters the convs of wavenet() read (wavenet_model.py:59-119,
 * 125-171): repacks the banks into per-workgroup LDS images and uploads them.  May be called again
 * after the parameters change. */
int wn_load_weights(wn_handle* h, const wn_weight_ptrs* w);

/* DilatedQueue.reset() for every layer (wavenet_modules.py:74-77, called at wavenet_model.py:250-251):
 * zeroes all queue rings of all streams and rewinds queue time to 0.  Asynchronous on hip_stream. */
int wn_reset(wn_handle* h, void* hip_stream);

/* The priming loop + hot loop of generate_fast (wavenet_model.py:259-311): ONE persistent kernel
 * launch for the whole job.  Does NOT reset the queues: call wn_reset first for a fresh
 * generate_fast; call again with n_given = 1 and first_samples = the last generated index to continue
 * a stream (used for progress callbacks).  Asynchronous: returns after enqueueing on hip_stream. */
int wn_generate(wn_handle* h, const wn_generate_args* args);

/* Blocks until the last wn_generate on this handle finished; returns its outcome
 * (WN_OK / WN_E_TIMEOUT / WN_E_HIP).  out_idx is valid afterwards. */
int wn_wait(wn_handle* h);

int wn_get_info(wn_handle* h, wn_info* out);

/* Copies the live queue of layer `layer`, stream `stream` into host memory laid out like the reference's
 * DilatedQueue.data after the same number of pushes: (R, (k-1)*d+1) row-major fp32, plus in_pos/out_pos
 * (wavenet_modules.py:43-57).  For tests and for facade code that inspects model.dilated_queues. */
int wn_export_queue(wn_handle* h, int32_t layer, int32_t stream, float* host_data, int32_t* in_pos, int32_t* out_pos);

/* WaveNetModel.forward() (wavenet_model.py:186-196) for one-hot inputs given as class indices: `indices` is a DEVICE
 * pointer to int32 [N][L]; writes fp32 logits [N*output_length][classes] (row = n*output_length + t, like the reference's
 * transpose+view at :194-196) to the DEVICE pointer `logits`.  The dilated-conv stack runs as fp32 matrix-core GEMMs
 * (csrc/wn_forward.h).  Asynchronous on hip_stream.  Clips shorter than receptive_field + output_length - 1 are served too: the
 * reference left-pads the layers' inputs with zero activations where their length is not a multiple of the dilation
 * (wavenet_modules.py:24-27) and the kernels read the tap x(t - d) as zero on exactly those rows
 * (round 4; the same holds for wn_train_forward / wn_train_backward: pad zeros carry no gradient).  WN_E_UNSUPPORTED for clip
 * lengths at which the reference itself has no defined result -- a layer left without an output position, the skip path's
 * un-dilation quirk at a per-row length of 1 (Appendix A item 17), fewer than output_length final positions; the message names
 * the layer --, for kernel_size != 2 and for channel counts that are not multiples of 32 (after zero padding): callers use the
 * torch path for those, which reproduces what the reference does there. */
int wn_forward(wn_handle* h, const int32_t* indices, int64_t N, int64_t L, int64_t output_length, float* logits, void* hip_stream);

/* Operand precision of wn_forward, wn_train_forward and the products of wn_train_backward (activation gradients AND weight
 * gradients; the one-hot product of start_conv stays fp32):
 * 0 = fp32 matrix-core GEMMs (default: equals the reference's fp32 graph to rounding), 1 = bf16 operands with fp32
 * accumulation (residual stream, skip sum and all accumulators stay fp32; the training step keeps the activations that only
 * ever feed bf16 operands -- z, tanh, sigmoid, [dF|dG] -- in bf16, which is the same rounding taken once at the store, and
 * since round 5 the skip convs' share of dz (one product per block of layers, added to the residual conv's share by the gate
 * derivative) as well: one more rounding point, carried by the step's oracle, oracle/bf16_step.py;
 * logits differ from the fp32 path at the 1e-2 level of their scale, gradients by a few per cent in norm -- mostly sign
 * flips of ReLU masks).  WN_E_UNSUPPORTED unless R, D, S and E are multiples of 64. */
int wn_set_forward_precision(wn_handle* h, int32_t bf16);

/* The priming loop of generate_fast (wavenet_model.py:259-269) for ALL given samples at once: `first_samples` is a DEVICE
 * pointer to int32 [n_streams][row_stride]; the first n_prime (= n_given - 1) samples of every stream are evaluated teacher
 * forced with the forward GEMM kernels (no skip / head work: the reference discards those outputs) and the queues are left
 * exactly as n_prime single evaluations would leave them.  Needs freshly reset queues (wn_reset); follow with
 * wn_generate(n_given = 1, first_samples = the last given sample).  WN_E_UNSUPPORTED under the same shape limits as
 * wn_forward: callers then prime through wn_generate.  Asynchronous on hip_stream. */
int wn_prime(wn_handle* h, const int32_t* first_samples, int64_t n_prime, int64_t row_stride, void* hip_stream);

/* ---- training step (WavenetTrainer.train, wavenet_training.py:58-107: output = model(x); loss = cross_entropy; loss.backward()) ----
 * The dilated-conv stack's forward AND backward run as fp32 matrix-core GEMMs; the caller owns the parameters as ONE flat
 * fp32 DEVICE array in the packed layout described by wn_train_layout (each reference parameter appears exactly once, so
 * any element-wise optimiser can step on the flat array directly), plus a same-shaped gradient array.  The loss (and its
 * gradient w.r.t. the logits) stays with the caller -- the reference computes it with F.cross_entropy (wavenet_training.py:83).
 * Offsets are in floats.  Packed layouts (NL layers, R/D/S/E/C channel counts, k = 2 taps):
 *   fg    [NL][2R][2D]  row = tap*R + r (tap 0 = x[t-d], tap 1 = x[t]); column = 64*(ch/32) + 32*gate + ch%32, gate 0 = filter
 *                       (filter_convs.l.weight[ch][r][tap], gate_convs.l.weight[ch][r][tap])       bfg [NL][2D] same columns
 *   res   [NL][D][R]    residual_convs.l.weight[r][d][0] transposed                                 bres [NL][R]
 *   skip  [NL][D][S]    skip_convs.l.weight[s][d][0] transposed                                     bskip [NL][S]
 *   bskip_total [S]     scratch (derived; its gradient is 0)
 *   w1 [S][E], b1 [E]   end_conv_1 transposed;   w2 [E][C], b2 [C]   end_conv_2 transposed
 *   start_t [C][R], start_b [R]   start_conv.weight[r][c][0] transposed
 * Bias sections exist (and are zero) also when the model has bias=False; their gradients are then left at 0. */
typedef struct wn_train_layout {
    int64_t total;
    int64_t fg, bfg, res, bres, skip, bskip, bskip_total, w1, b1, w2, b2, start_t, start_b;
} wn_train_layout;

int wn_train_get_layout(wn_handle* h, wn_train_layout* out);

/* Copies the parameters last given to wn_load_weights into `params` (DEVICE, layout.total floats). */
int wn_train_export_params(wn_handle* h, float* params, void* hip_stream);

/* The parameters' own tensors -- the reference's Conv1d layouts (out, in, k), one allocation per nn.Parameter (wavenet_model.py:59-119) -- as the
 * native training step takes and returns them (ABI 5).  filter_w .. skip_b: HOST arrays of n_layers DEVICE pointers (layer order); the others single
 * DEVICE pointers; the bias members are NULL for a model without stack biases. */
typedef struct wn_train_tensors {
    int32_t n_layers;          /* layers * blocks */
    int32_t reserved;          /* 0 */
    void* const* filter_w;     /* [n_layers] (D, R, 2) */
    void* const* gate_w;       /* [n_layers] (D, R, 2) */
    void* const* res_w;        /* [n_layers] (R, D, 1) */
    void* const* skip_w;       /* [n_layers] (S, D, 1) */
    void* const* filter_b;     /* [n_layers] (D) or NULL */
    void* const* gate_b;
    void* const* res_b;        /* (R) */
    void* const* skip_b;       /* (S) */
    void* start_w;             /* (R, classes, 1) */
    void* start_b;             /* (R) or NULL */
    void* end1_w;              /* (E, S, 1) */
    void* end1_b;
    void* end2_w;              /* (classes, E, 1) */
    void* end2_b;
} wn_train_tensors;

/* params (DEVICE, layout.total floats, overwritten) = the flat GEMM layout of the tensors: what a training step does with its nn.Parameters before
 * wn_train_forward, in a handful of launches (the tensors' addresses travel in the kernel arguments) instead of ~150 torch view / stack / copy
 * launches.  Every pointer the model's shape calls for must be set. */
int wn_train_pack(wn_handle* h, const wn_train_tensors* tensors, float* params, void* hip_stream);

/* The inverse, for gradients: grads (DEVICE, layout.total floats: what wn_train_backward wrote) -> one gradient tensor per parameter, in the
 * parameter's own layout.  A NULL pointer skips that tensor (the last layer's residual conv never reaches the loss: its .grad stays None upstream). */
int wn_train_unpack_grads(wn_handle* h, const float* grads, const wn_train_tensors* tensors, void* hip_stream);

/* on != 0: weight and bias gradients are bit-reproducible from run to run -- the row splits of every weight-gradient product store their partial
 * tiles in a workspace and a second kernel adds them in order, instead of fp32 atomics (default: off, or WN_DETERMINISTIC=1 in the environment at
 * wn_create; costs a write and a read of at most 67 MB of partial tiles per product).  The forward and the loss are deterministic either way. */
int wn_train_set_deterministic(wn_handle* h, int32_t on);

/* model(x) for training: like wn_forward (fp32) but reads the parameters from `params` and keeps every layer's input, gate
 * activations and the head's intermediates in a workspace owned by the handle for the following wn_train_backward. */
int wn_train_forward(wn_handle* h, const float* params, const int32_t* indices, int64_t N, int64_t L, int64_t output_length,
                     float* logits, void* hip_stream);

/* loss.backward(): `dlogits` [N*output_length][classes] (DEVICE) is dLoss/dlogits for the logits of the last
 * wn_train_forward on this handle; writes dLoss/dparams into `grads` (DEVICE, layout.total floats, overwritten).
 * Sums over rows are accumulated with fp32 atomics: results are reproducible to rounding, not bit for bit -- unless wn_train_set_deterministic. */
int wn_train_backward(wn_handle* h, const float* params, const float* dlogits, float* grads, void* hip_stream);

/* loss = F.cross_entropy(logits, targets) (mean over the M rows) and dLoss/dlogits in ONE pass over the logits -- the loss of the
 * training step (wavenet_training.py:69-70; torch runs log_softmax, nll_loss and their two backward kernels).  logits [M][classes]
 * fp32, targets [M] int64 class indices (torch's dtype), loss ONE float, dlogits [M][classes] or NULL: all DEVICE pointers.  Per
 * row: log-sum-exp in fp32; the mean is accumulated in fp64 in a fixed order (bit-reproducible).  A target outside [0, classes)
 * makes the row's loss NaN (torch raises a device assert).  WN_E_UNSUPPORTED unless classes == 256. */
int wn_train_loss(wn_handle* h, const float* logits, const int64_t* targets, int64_t M, float* loss, float* dlogits, void* hip_stream);

/* The optimiser half of the training step (wavenet_training.py:72-77: optional torch.nn.utils.clip_grad_norm over all parameters, then
 * optimizer.step() -- optim.Adam by default, :19-36) for a SET of tensors in a handful of launches (torch: ~130 multi-tensor launches on
 * config 5's 205 parameters): one pass for the gradients' total 2-norm, one pass that scales the gradients by min(1, max_norm / (norm + 1e-6)),
 * applies weight decay and takes the Adam step -- torch.optim.Adam's formulas operation for operation (csrc/wn_optim.h).  No handle:
 * params / grads / exp_avg / exp_avg_sq are HOST arrays of n_tensors DEVICE pointers to fp32 tensors of sizes[i] elements each (the caller's
 * nn.Parameters, their .grad and the optimiser's state; nothing is copied or kept); scratch: 8 bytes of DEVICE memory; total_norm: DEVICE
 * float or NULL.  step: the 1-based count of this step (bias corrections).  Asynchronous on hip_stream. */
typedef struct wn_adam_args {
    int32_t n_tensors;
    int32_t device_id;
    const int64_t* sizes;
    void* const* params;
    void* const* grads;
    void* const* exp_avg;
    void* const* exp_avg_sq;
    double lr, beta1, beta2, eps, weight_decay;   /* doubles, like the Python floats torch derives its fp32 scalars from (1 - beta2 is formed in double) */
    double max_grad_norm;  /* <= 0: no clipping (the reference's gradient_clipping=None) */
    int64_t step;
    float* total_norm;
    void* scratch;
    void* hip_stream;
    int64_t flags;         /* 0, or WN_ADAM_* below (ABI 5) */
} wn_adam_args;
/* Several parameter groups clipped TOGETHER (clip_grad_norm_ over model.parameters() in front of an optimiser with more than one group): add every
 * group's sum of squares into `scratch` first -- NORM_ONLY: no update is made; NORM_KEEP: `scratch` is not zeroed first (second group on) --, then step
 * every group with NORM_GIVEN: `scratch` already holds the sum of squares of all gradients, no norm pass.  One group: flags = 0 does both passes. */
#define WN_ADAM_NORM_ONLY  1
#define WN_ADAM_NORM_KEEP  2
#define WN_ADAM_NORM_GIVEN 4
int wn_adam_step(const wn_adam_args* args);

/* Diagnostics: record wall-clock stamps (100 MHz ticks) for the first n_items (evaluation, stream) steps of every
 * workgroup during the NEXT wn_generate: 8 slots per step -- 0 start, 1 input staged, 2 x' published, 3 done,
 * 4 filter/gate sums ready, 5 z staged, 6-7 unused -- then read them back as int64 [n_workgroups][n_items][8].  Used by tools/profile_chain.py. */
int wn_profile_next(wn_handle* h, int32_t n_items);
int wn_profile_read(wn_handle* h, int64_t* host_out, int64_t capacity);

/* Thread-local message for the most recent failing call on this thread ("" if none). */
const char* wn_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* WN_ABI_H */
