/*
 * imgxf.h — C-ABI of libimgxf.so: MI355X (gfx950) HIP kernels for the per-pixel
 * transform hot path of aaryaamoharir/ImageTransformations.
 *
 * The reference has no FFI of its own: its boundary is a set of Python functions
 * (`apply_<x>(img: PIL.Image, params) -> PIL.Image`, /root/reference/transformation.py:173-354,
 * and `TransformationPool.<x>`, /root/reference/pipenline/cifar_image_transformations.py:37-129)
 * whose bodies call Pillow / OpenCV / SciPy / NumPy C kernels.  Each entry point below
 * replaces ONE of those third-party calls; the call site it replaces is cited.  The
 * Python facade (imagetransformations_amd/transformation.py) binds these with ctypes
 * exactly as INTEGRATION.md shows.
 *
 * Conventions
 *   - every function returns IMGXF_OK (0), a negative imgxf error, or a positive hipError_t;
 *     nothing throws; nothing allocates or synchronises unless its comment says so;
 *   - all pixel pointers are DEVICE pointers (e.g. torch tensor .data_ptr()); the caller owns
 *     every buffer; `stream` is a hipStream_t (NULL = the default stream); calls are ordered
 *     by the stream only and are re-entrant;
 *   - images are interleaved (HWC) batches described by imgxf_view; strides are in BYTES;
 *     src and dst must not overlap unless the comment says in-place is allowed;
 *   - small host-side parameter arrays (kernels, matrices, colours) are HOST pointers that
 *     are copied into the kernel arguments before the call returns.
 */
#ifndef IMGXF_H
#define IMGXF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMGXF_VERSION 100 /* 0.1.0 */

enum {
    IMGXF_OK = 0,
    IMGXF_ERR_NULL = -1,        /* a required pointer is NULL */
    IMGXF_ERR_SHAPE = -2,       /* n/h/w/c or strides inconsistent between views */
    IMGXF_ERR_ARG = -3,         /* scalar argument out of range */
    IMGXF_ERR_UNSUPPORTED = -4, /* valid request this build has no kernel for */
    IMGXF_ERR_WORKSPACE = -5,   /* workspace too small */
    IMGXF_ERR_NO_DEVICE = -6    /* no gfx950-compatible device / code object */
};

/* A batch of n interleaved frames: pixel (f,y,x,ch) lives at
 * data + f*frame_stride + y*row_stride + (x*c + ch)*elem_size. */
typedef struct imgxf_view {
    void*   data;
    int32_t n, h, w, c;
    int64_t row_stride;
    int64_t frame_stride;
} imgxf_view;

enum { IMGXF_BORDER_REFLECT_101 = 0, /* gfedcb|abcdefgh|gfedcba (OpenCV default) */
       IMGXF_BORDER_REFLECT = 1      /* dcba|abcd|dcba (SciPy ndimage 'reflect') */ };

enum { IMGXF_FILTER_NEAREST = 0, IMGXF_FILTER_BILINEAR = 1, IMGXF_FILTER_BICUBIC = 2 };

enum { IMGXF_SOBEL_X_WRAP = 0,    /* scipy.ndimage.sobel(u8, axis=-1): result mod 256 */
       IMGXF_SOBEL_Y_WRAP = 1,    /* axis=0 */
       IMGXF_SOBEL_MAGNITUDE = 2  /* sat_u8(rint(sqrt(Gx^2+Gy^2))) — benchmark configs[2] */ };

int         imgxf_version(void);
const char* imgxf_strerror(int code);
/* Number of visible HIP devices whose architecture this library was built for (>=0),
 * or a negative error.  Does not create a context on any device. */
int         imgxf_device_count(void);
/* The IMGXF_* tuning / routing environment variables are read once, at the first call that needs
 * one.  This re-reads them (not thread-safe against concurrent launches; for tests and A/B tools
 * that change a knob inside one process).  No reference counterpart (build infrastructure). */
int         imgxf_reload_knobs(void);
/* Measurement aid for bench.py (no reference counterpart): one wave spins for `ticks_100mhz` ticks
 * of the constant 100 MHz counter and writes {delta s_memtime, delta s_memrealtime} to the two
 * uint64 at `out2_u64` (device memory); shader clock in MHz = 100 * out[0] / out[1].  Launch it on
 * a side stream while the kernels of interest run to see the clock they are granted. */
int         imgxf_probe_sclk(void* out2_u64, unsigned int ticks_100mhz, void* stream);

/* ---- a1: cv2.GaussianBlur(img,(k,k),sigma)  transformation.py:249 -------------------
 * Separable Gaussian, BORDER_REFLECT_101, fp32 accumulate, round-half-even, saturate.
 * c in {1,3,4}; ksize odd, 1..31.  sigma<=0 as OpenCV: the binomial table for ksize <= 7,
 * sigma = 0.3*((k-1)*0.5-1)+0.8 beyond.
 * `dst_f32` (optional, may be NULL): if given, a float view of the same n,h,w,c that
 * receives the pre-quantisation fp32 values (diagnostic; used by the parity tests). */
int imgxf_gaussian_u8(const imgxf_view* src, const imgxf_view* dst, int ksize, double sigma,
                      const imgxf_view* dst_f32, void* stream);

/* Generic separable correlation with host-given fp32 taps (kx along x, ky along y). */
int imgxf_sepconv_u8(const imgxf_view* src, const imgxf_view* dst, const float* kx, int nkx,
                     const float* ky, int nky, int border, const imgxf_view* dst_f32,
                     void* stream);

/* The same filter the way OpenCV >= 4 most likely evaluates it for 8-bit images (its fixed-point
 * path; restated, cv2 is not installed — PARITY UNPINNED): taps quantised to 8 fractional bits
 * (getGaussianKernelFixedPoint_ED: error-diffused rounding, centre = 256 - rest), 8.8 rows,
 * 16.16 columns, (v + 2^15) >> 16.  Not the contract path (that is the float definition above);
 * offered because the reference's own JPEG outputs sit closer to it.
 * sepconv_fixed: host-given 8.8 integer taps, each axis summing to <= 256. */
int imgxf_gaussian_cv_fixed_u8(const imgxf_view* src, const imgxf_view* dst, int ksize,
                               double sigma, void* stream);
int imgxf_sepconv_fixed_u8(const imgxf_view* src, const imgxf_view* dst, const uint16_t* kx,
                           int nkx, const uint16_t* ky, int nky, int border, void* stream);

/* ---- a5: cv2.filter2D(img,-1,kernel)  cifar_image_transformations.py:118 ------------
 * Dense kh x kw correlation, centre anchor, fp32 accumulate, round-half-even, saturate.
 * kernel: HOST pointer, row-major kh*kw floats; kh,kw odd, <= 15. */
int imgxf_conv2d_u8(const imgxf_view* src, const imgxf_view* dst, const float* kernel, int kh,
                    int kw, int border, void* stream);

/* ---- a4: scipy.ndimage.sobel(gray_u8)  transformation.py:339 ------------------------
 * src, dst: c == 1.  variant: IMGXF_SOBEL_*.  Border: SciPy 'reflect'. */
int imgxf_sobel_u8(const imgxf_view* src, const imgxf_view* dst, int variant, void* stream);
/* Fused benchmark configs[2]: RGB(c==3) -> L (Pillow weights) -> Gx,Gy -> magnitude -> u8 (c==1). */
int imgxf_rgb_sobel_mag_u8(const imgxf_view* src, const imgxf_view* dst, void* stream);
/* Same fusion for every variant of imgxf_sobel_u8: RGB in, Pillow L on the fly, one u8 plane out
 * (apply_background_change's edge image without materialising L, transformation.py:336-339). */
int imgxf_rgb_sobel_u8(const imgxf_view* src, const imgxf_view* dst, int variant, void* stream);

/* ---- a2/a2'/shear: Image.transform(size, AFFINE, m, resample, fillcolor) -------------
 * transformation.py:200 (rotate -> NEAREST), :217-224 (shear -> BICUBIC); BILINEAR is
 * benchmark configs[3].  m[6]: HOST pointer, destination->source matrix exactly as Pillow
 * takes it.  dst gives the output size.  fill[4]: HOST pointer (per channel), NULL = zeros.
 * NEAREST reproduces libImaging affine_fixed (16.16) bit-exactly when m[1]!=0 or m[3]!=0;
 * pure scale/translate NEAREST matrices need imgxf_affine_scale_nearest_u8.
 * precise != 0: coordinates and interpolation in fp64 (bit-exact with Pillow);
 * precise == 0: fp64 coordinates, fp32 interpolation (<=1e-5 relative before truncation).
 * `dst_f32` (optional, may be NULL; BILINEAR/BICUBIC only): float view of dst's n,h,w,c that
 * receives the pre-truncation interpolated values (diagnostic; used by the parity tests). */
int imgxf_affine_u8(const imgxf_view* src, const imgxf_view* dst, const double* m, int filter,
                    const uint8_t* fill, int precise, const imgxf_view* dst_f32, void* stream);
/* libImaging ImagingScaleAffine (NEAREST with m[1]==m[3]==0): source indices are walked on
 * one device lane per axis by the same repeated double additions Pillow performs, into
 * `workspace` (device, 4-byte aligned, >= 4*(dst->w + dst->h + 2) bytes). */
int imgxf_affine_scale_nearest_u8(const imgxf_view* src, const imgxf_view* dst, const double* m,
                                  const uint8_t* fill, void* workspace, size_t workspace_bytes,
                                  void* stream);

/* ---- a3: img.resize((nw,nh), LANCZOS)  transformation.py:179 -------------------------
 * Two-pass integer resample (22-bit coefficients computed on the host in double exactly as
 * libImaging precompute_coeffs/normalize_coeffs_8bpc).  A plan owns the device coefficient
 * tables and the uint8 intermediate for up to `max_frames` frames; create/destroy allocate,
 * synchronise and must not be called inside stream capture; the resize call only launches.
 * A plan may be in use on one stream at a time (its intermediate is shared). */
typedef struct imgxf_lanczos_plan imgxf_lanczos_plan;
int imgxf_lanczos_plan_create(imgxf_lanczos_plan** plan, int in_h, int in_w, int out_h,
                              int out_w, int c, int max_frames);
int imgxf_lanczos_plan_destroy(imgxf_lanczos_plan* plan);
int imgxf_resize_lanczos_u8(const imgxf_lanczos_plan* plan, const imgxf_view* src,
                            const imgxf_view* dst, void* stream);
/* The same machinery with Resample.c's other filters — Image.resize's default BICUBIC is what
 * rand_crop uses (fall_2025/transformations_code:43-48).  `filter` takes Pillow's Resampling
 * values; the plan is run and destroyed with the two calls above. */
enum { IMGXF_RESAMPLE_LANCZOS = 1, IMGXF_RESAMPLE_BILINEAR = 2, IMGXF_RESAMPLE_BICUBIC = 3,
       IMGXF_RESAMPLE_BOX = 4, IMGXF_RESAMPLE_HAMMING = 5 };
int imgxf_resample_plan_create(imgxf_lanczos_plan** plan, int in_h, int in_w, int out_h,
                               int out_w, int c, int max_frames, int filter);
/* Resize + crop in one: the plan produces only rows [wy, wy+wh) x columns [wx, wx+ww) of the
 * out_h x out_w result (dst of the resize call is wh x ww) and filters only the source rows and
 * output columns that window needs — apply_scale's centre crop (transformation.py:182-187)
 * without computing the pixels it throws away.  Needs out_w != in_w and out_h != in_h. */
int imgxf_resample_plan_create_window(imgxf_lanczos_plan** plan, int in_h, int in_w, int out_h,
                                      int out_w, int c, int max_frames, int filter,
                                      int wx, int wy, int ww, int wh);
/* Stream-safe form: with max_frames == 0 a plan holds only its immutable coefficient tables; the
 * uint8 H -> V intermediate ([n][rows the window's taps touch][out_w][c] bytes, 0 for single-pass
 * plans) is a caller-provided, stream-ordered workspace, so one plan may run on any number of
 * streams and batch sizes at once and the call neither allocates nor frees (graph-capture safe).
 * Replaces the shared-intermediate contract of imgxf_resize_lanczos_u8 above. */
int imgxf_resample_workspace_bytes(const imgxf_lanczos_plan* plan, int n, size_t* bytes);
int imgxf_resample_ws_u8(const imgxf_lanczos_plan* plan, const imgxf_view* src, const imgxf_view* dst,
                         void* workspace, size_t workspace_bytes, void* stream);
/* The workspace THIS call needs: 0 when the two views let the fused kernel run (see
 * imgxf_resample_plan_kernel), else imgxf_resample_workspace_bytes(plan, src->n). */
int imgxf_resample_workspace_bytes_for(const imgxf_lanczos_plan* plan, const imgxf_view* src,
                                       const imgxf_view* dst, size_t* bytes);
/* Which kernels a two-pass plan runs on 4-byte aligned views: *ksteps = 0 -> the H and V vector
 * kernels through the intermediate; 1..3 -> both passes fused on the i8 matrix cores
 * (csrc/resample_mfma.inc, no intermediate traffic), the value being the 32-byte k-steps of its
 * horizontal product.  Same bytes either way; IMGXF_RESAMPLE_NO_MFMA=1 forces the former. */
int imgxf_resample_plan_kernel(const imgxf_lanczos_plan* plan, int* ksteps);

/* ---- a6: elementwise colour maps ------------------------------------------------------*/
/* Pillow convert('L') transformation.py:336: (19595R+38470G+7471B+0x8000)>>16. src c in {3,4}, dst c==1 */
int imgxf_rgb2l_u8(const imgxf_view* src, const imgxf_view* dst, void* stream);
/* cv2.convertScaleAbs(img, alpha, beta) transformation.py:207: sat_u8(rint(|alpha*p+beta|)). In-place ok. */
int imgxf_scale_abs_u8(const imgxf_view* src, const imgxf_view* dst, float alpha, float beta,
                       void* stream);
/* Image.blend(im1, im2, alpha) (libImaging Blend.c) transformation.py:267,354:
 * f32 `p1 + alpha*(p2-p1)`; 0<=alpha<=1 truncates, else clip then truncate.
 * im2 == NULL: the second image is the solid colour `color2[c]` (HOST pointer).
 * im1 == NULL: the first image is the solid colour `color1[c]`.  In-place ok. */
int imgxf_blend_u8(const imgxf_view* im1, const uint8_t* color1, const imgxf_view* im2,
                   const uint8_t* color2, const imgxf_view* dst, float alpha, void* stream);
/* transformation.py:275-278: clip(f32(p)+noise,0,255) truncated; noise: float view, same n,h,w,c. */
int imgxf_add_noise_u8(const imgxf_view* src, const imgxf_view* noise_f32, const imgxf_view* dst,
                       void* stream);
/* The same step with the noise tensor GENERATED ON THE DEVICE (opt-in, IMGXF_NOISE_RNG=device in the facade): the host's
 * np.random.normal(0, sigma, shape).astype(f32) at transformation.py:274 is replaced by Philox4x32-10 (key = seed,
 * counter = offset / 4 + element index / 4) + Box-Muller in fp32, scaled by `sigma` (= noise_std * 255); then
 * clip(f32(p) + noise, 0, 255) truncated as above.  A different random stream than NumPy's MT19937: distribution-level
 * parity only.  `offset` (a multiple of 4) numbers the first normal of this call, so frames of one
 * logical batch can be processed in several calls with identical results.  In-place ok. */
int imgxf_add_noise_philox_u8(const imgxf_view* src, const imgxf_view* dst, float sigma, uint64_t seed,
                              uint64_t offset, void* stream);
/* The raw uint32 stream behind it (tests: Random123 known-answer vectors): count (multiple of 4) values to dst_u32. */
int imgxf_philox4x32_u32(void* dst_u32, int64_t count, uint64_t seed, uint64_t offset, void* stream);
/* cv2.cvtColor channel permutations (RGB2BGR, RGBA2RGB, ...) transformation.py:206,233-235,252:
 * dst[..., j] = src[..., perm[j]] for j < dst->c.  perm: HOST pointer. */
int imgxf_permute_u8(const imgxf_view* src, const imgxf_view* dst, const int32_t* perm,
                     void* stream);
/* Image.composite(im1, im2, mask) transformation.py:344 for a 0/255 mask (c==1): mask ? im1 : im2. */
int imgxf_composite_u8(const imgxf_view* im1, const imgxf_view* im2, const imgxf_view* mask,
                       const imgxf_view* dst, void* stream);
/* Image.composite(im1, Image.new(mode, size, colour), mask) without materialising the constant image
 * (transformation.py:333,344). */
int imgxf_composite_const_u8(const imgxf_view* im1, const uint8_t* colour, const imgxf_view* mask,
                             const imgxf_view* dst, void* stream);

/* ---- Image.filter(ImageFilter.Kernel((3,3), kernel, scale, offset)) — libImaging ImagingFilter3x3
 * (ImageFilter.SMOOTH behind ImageEnhance.Sharpness, cifar_image_transformations.py:95-99):
 * float32 coefficients kernel9[i]/scale (HOST pointer, Pillow's order: first triple = row y+1),
 * ss = offset + 0.5 then one (a*k0 + b*k1) + c*k2 per row, clip8 truncation; the one-pixel
 * frame of the image is copied from the input.  Any c. */
int imgxf_filter3x3_u8(const imgxf_view* src, const imgxf_view* dst, const float* kernel9,
                       float scale, float offset, void* stream);

/* ---- TransformationPool noise members, device half (cifar_image_transformations.py:39-70) ------
 * The random draws stay on the host with NumPy's generator, exactly as the reference makes them;
 * these apply them.  All float views are float64 (8-byte aligned).
 * add_noise_f64:  dst = trunc(clip(f64(f32(p)) + noise, 0, 255))                   (:45-47)
 * shot_noise:     dst = trunc(clip(counts / lambda * 255.0, 0, 255)), counts = Poisson draws (:68-69)
 * impulse_noise:  mask [n,h,w,1]: mask < lo -> 0, mask > hi -> 255, else src      (:57-58) */
int imgxf_add_noise_f64_u8(const imgxf_view* src, const imgxf_view* noise_f64, const imgxf_view* dst, void* stream);
int imgxf_shot_noise_u8(const imgxf_view* counts_f64, double lambda, const imgxf_view* dst, void* stream);
int imgxf_impulse_noise_u8(const imgxf_view* src, const imgxf_view* mask_f64, double lo, double hi,
                           const imgxf_view* dst, void* stream);

/* ---- AugMix point ops and histograms  fall_2025/AugMix.py:31,36,37; Initial_Experiments.py:95-113
 * lut:       dst = lut[channel][src], lut = host table of c*256 bytes (ImageOps.posterize /
 *            solarize / any Image.point table); copied into the launch, no device allocation.
 * equalize:  ImageOps.equalize per frame and channel (histogram -> table -> map), all on the
 *            device.  workspace: >= n*c*256*5 bytes, 4-byte aligned.
 * channel_histogram: hist[n][c][256] uint32 (device, zeroed by the call) of an interleaved view. */
int imgxf_lut_u8(const imgxf_view* src, const imgxf_view* dst, const uint8_t* lut, void* stream);
int imgxf_equalize_u8(const imgxf_view* src, const imgxf_view* dst, void* workspace, size_t workspace_bytes,
                      void* stream);
int imgxf_channel_histogram_u8(const imgxf_view* src, uint32_t* hist, void* stream);

/* ---- TransformationPool.histogram_equalization  cifar_image_transformations.py:122-129 -------
 * cv2.cvtColor(RGB2YUV / YUV2RGB) for 8-bit images (integer BT.601, yuv_shift 14) and
 * cv2.equalizeHist applied to one channel of an interleaved view.  PARITY UNPINNED: OpenCV is not
 * installed in the build container; the arithmetic follows OpenCV's integer definitions.
 * workspace of equalize_hist_cv: >= n*c*256*5 bytes, 4-byte aligned. */
int imgxf_rgb2yuv_u8(const imgxf_view* src, const imgxf_view* dst, void* stream);
int imgxf_yuv2rgb_u8(const imgxf_view* src, const imgxf_view* dst, void* stream);
int imgxf_equalize_hist_cv_u8(const imgxf_view* src, const imgxf_view* dst, int channel, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---- ImageFilter.BoxBlur / ImageFilter.GaussianBlur — libImaging BoxBlur.c --------------------
 * (TransformationPool.defocus_blur, cifar_image_transformations.py:72-77.)  `passes` box passes
 * along x then along y, each in exact uint32 arithmetic with replicated edges and a uint8
 * intermediate.  workspace: device scratch of n*h*w*c bytes (needed when more than one pass
 * runs).  gaussian_blur_pil: box radius from Pillow's _gaussian_blur_radius, 3 passes. */
int imgxf_box_blur_u8(const imgxf_view* src, const imgxf_view* dst, float xradius, float yradius,
                      int passes, void* workspace, size_t workspace_bytes, void* stream);
int imgxf_gaussian_blur_pil_u8(const imgxf_view* src, const imgxf_view* dst, float radius,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ---- ImageEnhance.Color / .Contrast ---------------------------------
 * pipenline/cifar_image_transformations.py:81-85,102-106.  Both are Image.blend(degenerate,
 * image, factor) with the Blend.c float semantics above.
 * Color: degenerate = convert('L') replicated to RGB (fused per pixel).  c == 3.
 * Contrast: degenerate = solid int(mean(L) + 0.5) per frame; `sums` is device scratch of n
 * uint64 (zeroed by the call) that receives every frame's sum of L.  c in {1,3}. */
int imgxf_enhance_color_u8(const imgxf_view* src, const imgxf_view* dst, float factor, void* stream);
int imgxf_enhance_contrast_u8(const imgxf_view* src, const imgxf_view* dst, float factor,
                              uint64_t* sums, void* stream);

/* ---- crop / paste / fill (Image.crop, Image.paste, Image.new) transformation.py:187-193,287-305 */
/* Fill every pixel of dst with color[c] (HOST pointer). */
int imgxf_fill_u8(const imgxf_view* dst, const uint8_t* color, void* stream);
/* Copy the rectangle (sx,sy,rw,rh) of src to (dx,dy) of dst for every frame (same n, c). */
int imgxf_copy_rect_u8(const imgxf_view* src, const imgxf_view* dst, int sx, int sy, int dx,
                       int dy, int rw, int rh, void* stream);
/* apply_translation (transformation.py:284-307: Image.new(black) + crop + paste) in one pass:
 * dst(x, y) = src(x - dx, y - dy) where that pixel exists, else fill[c] (HOST pointer).  Same n,h,w,c. */
int imgxf_translate_u8(const imgxf_view* src, const imgxf_view* dst, int dx, int dy,
                       const uint8_t* fill, void* stream);
/* Image.transpose(ROTATE_90/180/270) fast paths of Image.rotate (PIL/Image.py:2513-2521).
 * quarter_turns_ccw in {1,2,3}. */
int imgxf_rot90_u8(const imgxf_view* src, const imgxf_view* dst, int quarter_turns_ccw,
                   void* stream);
/* Image.transpose(FLIP_LEFT_RIGHT) (mode 0; vert_flip, fall_2025/transformations_code:39-41) or
 * FLIP_TOP_BOTTOM (mode 1).  Same geometry in and out. */
int imgxf_flip_u8(const imgxf_view* src, const imgxf_view* dst, int mode, void* stream);

/* ---- float-tensor corruption maps  pipenline/angellic.py:34-46, angellic2.py:47-50 -------*/
/* Unnormalised fp32 images in [0,1], any shape, `count` contiguous elements (device pointers,
 * 4-byte aligned; in-place allowed).  mode BRIGHTNESS: clamp(x + p0, 0, 1); CONTRAST:
 * clamp((x - 0.5)*p0 + 0.5, 0, 1); NOISE: clamp(x + (noise*p0 + p1), 0, 1) with `noise` the
 * caller's torch.randn_like draw (std = p0, mean = p1).  Same fp32 operations in the same order as
 * torch's eager kernels (bit-identical).  mask (optional, device, `count` bytes): 1 where the
 * value before the clamp lay in [0,1], i.e. where torch.clamp's backward passes the gradient. */
enum { IMGXF_F32_BRIGHTNESS = 0, IMGXF_F32_CONTRAST = 1, IMGXF_F32_NOISE = 2 };
int imgxf_f32_map(const float* src, const float* noise, float* dst, uint8_t* mask, int64_t count,
                  int mode, float p0, float p1, void* stream);

/* transforms.ToTensor() (+ transforms.Normalize(mean, std)): uint8 HWC frames -> float32
 * [n][c][h][w] planes at `dst` (device, contiguous), x/255 correctly rounded, then
 * (x - mean[c]) / std[c] in fp32 — bit-identical to torchvision's tensor arithmetic.  mean / std:
 * HOST float[c], both NULL = ToTensor only.  c in {1,3,4}. */
int imgxf_to_tensor_f32(const imgxf_view* src, float* dst, const float* mean, const float* std,
                        void* stream);

/* ---- perspective warp  fall_2025/transformations_code:54-66 -----------------------------*/
/* torchvision RandomPerspective on a float tensor for given coefficients: ToTensor (u8/255),
 * _perspective_grid + grid_sample(bilinear, padding zeros, align_corners=False) of the image and
 * a ones channel, img*mask + (1-mask)*0, ToPILImage (mul(255).byte()), all in fp32 in the
 * evaluation order of the torch CPU build.  coeffs: HOST float[8] (per_frame = 0, shared by
 * all frames) or float[n][8] (per_frame = 1), as returned by torchvision's
 * _get_perspective_coeffs (output pixel -> source).  src and dst have the same n, h, w, c
 * (c in {1,3,4}) and must not alias. */
int imgxf_perspective_bilinear_u8(const imgxf_view* src, const imgxf_view* dst,
                                  const float* coeffs, int per_frame, void* stream);

/* ---- the driver's save step  transformation.py:161-162 (`transformed.save(path)`: Pillow ->
 * libjpeg-turbo baseline JPEG, 4:2:0, islow DCT, no restart markers) ------------------------*/
typedef struct imgxf_jpeg_tables {
    uint16_t quant[2][64];      /* luminance / chrominance divisors 1..255, natural (row-major) order */
    uint16_t dc_code[2][16];    /* Huffman code of DC magnitude category 0..11 (table 0: Y, 1: Cb/Cr) */
    uint8_t  dc_len[2][16];
    uint16_t ac_code[2][256];   /* Huffman code of the AC symbol (run << 4) | size */
    uint8_t  ac_len[2][256];
} imgxf_jpeg_tables;
/* One complete JPEG file per frame of a c == 3 view: out + f*out_frame_stride holds `header`
 * (host bytes, SOI .. SOS as the caller built them for these tables and this size, <= 1024),
 * the entropy-coded segment, EOI; sizes[f] (device) = the file's length, or 0xFFFFFFFF when it
 * does not fit in out_frame_stride bytes (nothing usable is written for that frame).
 * Bit-identical to libjpeg(-turbo)'s output for the same tables.  workspace: device, 16-byte
 * aligned, >= imgxf_jpeg_workspace_bytes(n, h, w, out_frame_stride).  Limits: n <= 65535,
 * out_frame_stride <= 2^31, ceil(w/16)*ceil(h/16) < 349525 MCUs per frame (bit offsets are
 * 32-bit; 8K frames fit), else IMGXF_ERR_SHAPE / IMGXF_ERR_ARG. */
int imgxf_jpeg_workspace_bytes(int n, int h, int w, size_t out_frame_stride, size_t* bytes);
int imgxf_jpeg_encode_u8(const imgxf_view* src, const imgxf_jpeg_tables* tables, const uint8_t* header,
                         int header_bytes, uint8_t* out, size_t out_frame_stride, uint32_t* sizes,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---- mask stage of apply_background_change  transformation.py:340-341 -----------------*/
/* 256-bin histogram per frame of a c==1 view into hist[n][256] (uint32, device, zeroed by the call). */
int imgxf_histogram_u8(const imgxf_view* src, uint32_t* hist, void* stream);
/* mask = (src > np.percentile(src, q)) ? 255 : 0 per frame, threshold derived on the device
 * from `hist` (numpy 'linear' method).  thr_out: device double[n] (required), receives the
 * percentile of every frame. */
int imgxf_percentile_mask_u8(const imgxf_view* src, const uint32_t* hist, double q,
                             const imgxf_view* dst, double* thr_out, void* stream);
/* scipy.ndimage.binary_dilation(mask, iterations) with the 4-connected cross, border 0:
 * equals "L1 distance <= iterations" for this structuring element.  0/255 masks, c==1.
 * iterations in 1..16. */
int imgxf_dilate_cross_u8(const imgxf_view* src, const imgxf_view* dst, int iterations,
                          void* stream);

/* ---- f-4 (decode half): Image.open(path).convert("RGB")  transformation.py:83 -----------------------
 * Baseline / extended-sequential Huffman JPEG, 8 bit, 1 or 3 components (Y or YCbCr; 4:4:4, 4:2:2 h2v1, 4:2:0 h2v2),
 * decoded the way Pillow's libjpeg-turbo does with its defaults: jpeg_idct_islow, fancy (triangle) upsampling, the
 * 16-bit fixed-point YCbCr -> RGB tables, grayscale replicated — bit-identical pixels.  The HOST parses the markers,
 * removes the byte stuffing, splits the scan at RSTn markers and derives the lookup tables (imagetransformations_amd/
 * jpeg.py); the device does the entropy decoding (one thread per restart segment — a file without restart markers is one
 * segment: Huffman decoding is the serial direction), dequantisation + IDCT, upsampling and colour conversion.
 * All arrays are DEVICE pointers; offsets are in elements of the array they index. */
typedef struct imgxf_jpeg_dec_comp {
    int32_t h, v;               /* sampling factors (1 or 2) */
    int32_t dc_tab, ac_tab;     /* indices into luts[] (imgxf_jpeg_dec_lut) */
    int32_t quant;              /* index into quants[] (64 uint16 each, natural order) */
    int32_t blocks_x, blocks_y; /* allocated blocks: mcux * h, mcuy * v */
    int32_t dw, dh;             /* downsampled_width / height: ceil(width * h / hmax), ceil(height * v / vmax) */
    int64_t coef_off;           /* first int16 coefficient of the component in coefs[] ([blocks_y][blocks_x][64], ZIGZAG order: the stream's) */
    int64_t plane_off;          /* first byte of the component's sample plane in planes[] (pitch = 8 * blocks_x) */
} imgxf_jpeg_dec_comp;
typedef struct imgxf_jpeg_dec_image {
    int32_t width, height, ncomp, hmax, vmax, mcux, mcuy;
    int32_t restart_interval;   /* MCUs per segment (the whole scan when the file has no DRI) */
    int32_t seg_first, seg_count; /* this image's entries of seg_off[] (byte offset of each segment in scan[]) and seg_len[] */
    int32_t pad_;
    int64_t out_off;            /* first byte of the image's RGB pixels in out[] (row pitch = out_pitch bytes) */
    int64_t out_pitch;
    imgxf_jpeg_dec_comp comp[3];
} imgxf_jpeg_dec_image;
typedef struct imgxf_jpeg_dec_lut {
    uint16_t look[256];         /* 8-bit lookahead: (code length << 8) | symbol, 0 = the code is longer than 8 bits */
    int32_t  maxcode[18];       /* jdhuff.c: largest code of each length (index 1..16), -1 if none; [17] = sentinel */
    int32_t  valoff[17];        /* huffval index of the first code of each length minus that code */
    uint8_t  huffval[256];
} imgxf_jpeg_dec_lut;
/* Entropy decoding of n images: coefs[] must be ZERO on entry (only non-zero coefficients are written).  status[i]
 * (device int32, may be NULL) receives 0, or 1 when image i's stream held an impossible code / ran out of data. */
int imgxf_jpeg_decode_huffman(const uint8_t* scan, const int64_t* seg_off, const int32_t* seg_len,
                              const imgxf_jpeg_dec_image* images, int n, const imgxf_jpeg_dec_lut* luts,
                              int16_t* coefs, int32_t* status, void* stream);
/* HOST helper of the reader (no device work): walks the entropy-coded bytes of one file's scan from data[start], removes
 * the byte stuffing (FF 00 -> FF), splits at RSTn and stops at the first other marker (or at n; *ecs_end = that position).
 * At most max_segs segments are kept (the scan's ceil(MCUs / restart interval); later ones are skipped).  Segment k is
 * written to scan[] at seg_off[k] (16-byte aligned, taken from *scan_pos, which is advanced) with seg_len[k] bytes followed
 * by 16 .. 31 zero bytes.  IMGXF_ERR_WORKSPACE if scan_cap is too small.  Replaces the per-byte Python walk of
 * jpeg_decode._segments for the batched reader (load_data's Image.open, /root/reference/transformation.py:73-89). */
int imgxf_jpeg_unstuff_host(const uint8_t* data, size_t n, size_t start, uint8_t* scan, size_t scan_cap, size_t* scan_pos,
                            int64_t* seg_off, int32_t* seg_len, int max_segs, int* nsegs, size_t* ecs_end);
/* NumPy's legacy generator on the device (np.random.normal of apply_gaussian_noise, /root/reference/transformation.py:273-275):
 * the raw MT19937 state sequence — out[0 .. 623] = key (device pointer, the generator's current 624 state words), block b =
 * the state after b regenerations (mt19937_gen), (nblocks + 1) * 624 words in all.  One workgroup: the recurrence is
 * sequential in the block index.  Tempering, legacy_double and the polar method follow in imagetransformations_amd/numpy_stream.py. */
int imgxf_mt19937_blocks(const uint32_t* key, uint32_t* out, int64_t nblocks, void* stream);

/* The same sequence in parallel.  imgxf_mt19937_jump: out_keys[w * 624 ..] = the generator's canonical state (w + 1) * J words
 * after base_key[0 .. 623], for w = 0 .. n_out - 1, all in parallel; J = 624 * 2^k words is the stride whose jump polynomials
 * `coefs` holds ([n_out][2496] bytes: 19937 coefficients each, little-endian bits; imagetransformations_amd/mt19937_jump.npz,
 * written and checked by tools/make_mt_jump.py).  Word 0 of a jumped state is state only in its top bit.
 * imgxf_mt19937_stretches: workgroup m writes blocks m * blocks_per_stretch .. of the state sequence from keys[m]; total_blocks
 * in all.  Together they produce exactly what imgxf_mt19937_blocks produces. */
int imgxf_mt19937_jump(const uint32_t* base_key, uint32_t* out_keys, int n_out, const uint8_t* coefs, void* stream);
int imgxf_mt19937_stretches(const uint32_t* keys, uint32_t* out, int n_stretches, int64_t blocks_per_stretch, int64_t total_blocks,
                            void* stream);

/* NumPy's legacy_gauss over the word stream of imgxf_mt19937_blocks / _stretches (numpy/random/src/legacy/legacy-distributions.c
 * restated; imagetransformations_amd/numpy_stream.py): group g = four consecutive words -> (x1, x2, r2), accepted iff 0 < r2 < 1.
 * imgxf_np_accept writes the acceptance flags; with their inclusive prefix sum `rank`, imgxf_np_normals_f32 writes the float32
 * results of consecutive np.random.normal(0, scale, count) calls: accepted group k <= groups yields normals 2 (k - 1), 2 (k - 1) + 1
 * (f x2 then f x1) of n2; out[e + lead] = float(0.0 + scale * f x) with the scale of the request whose range (reqs: {int64 begin,
 * double scale} sorted by begin, positions counted with `lead` = 1 if a cached normal precedes) holds it.  Samples within `margin`
 * (relative) of a float32 rounding boundary are listed in risky[] (info[1] = how many; the host recomputes them with its libm);
 * info[0] = the index of the groups-th accepted group, xr[0 .. 1] = its (x1, r2), xr[2 + 2 s ..] = (x, r2) of risky sample s
 * (xr holds 2 + 2 risky_cap doubles). */
int imgxf_np_accept(const uint32_t* words, int64_t ngroups, uint8_t* acc, void* stream);
int imgxf_np_normals_f32(const uint32_t* words, int64_t ngroups, const int64_t* rank, int64_t groups, int64_t n2, int lead,
                         const void* reqs, int nreq, double margin, float* out, int64_t* info, int64_t* risky, int64_t risky_cap,
                         double* xr, void* stream);

/* Why a file is outside the reader's class, or damaged (status[] of imgxf_jpeg_layout_host; 0 = accepted). */
enum { IMGXF_JPEG_E_NOT_JPEG = 1,    /* no SOI */
       IMGXF_JPEG_E_MARKERS = 2,     /* damaged marker structure (also: SOS before SOF) */
       IMGXF_JPEG_E_PRECISION = 3,   /* samples are not 8 bits */
       IMGXF_JPEG_E_PROCESS = 4,     /* progressive, lossless or arithmetic coding */
       IMGXF_JPEG_E_COMPONENTS = 5,  /* neither 1 nor 3 components, or a non-interleaved scan */
       IMGXF_JPEG_E_SCAN_ORDER = 6,  /* the scan names an unknown component or not in frame order */
       IMGXF_JPEG_E_SAMPLING = 7,    /* sampling factors outside 1..2 */
       IMGXF_JPEG_E_CHROMA = 8,      /* chroma sampling other than 4:4:4, 4:2:2 (h2v1), 4:2:0 */
       IMGXF_JPEG_E_NO_QUANT = 9,    /* a component's quantisation table is missing */
       IMGXF_JPEG_E_NO_HUFF = 10,    /* a component's Huffman table is missing */
       IMGXF_JPEG_E_TRUNCATED = 11   /* the scan ends before its last restart segment */ };
/* HOST half of the reader for a batch of n files (no device work; csrc/jpeg_layout.hip): marker segments up to the scan
 * (jdmarker.c), image / component descriptors, quantisation tables in natural order, derived Huffman tables
 * (jpeg_make_d_derived_tbl; equal tables are shared), and the entropy-coded bytes laid out by imgxf_jpeg_unstuff_host.
 * Pass 1, scan == NULL: only the counts — *n_segs, *n_quants, *n_luts and *scan_bytes are (bounds on) what pass 2 needs.
 * Pass 2: images[n], luts[*n_luts], quants[*n_quants][64], scan[*scan_bytes], seg_off / seg_len[*n_segs] are filled,
 * *coef_total / *plane_total are the sizes of coefs[] (int16) and planes[] (bytes); out_off / out_pitch of the images are
 * left to the caller.  status[i] is 0 or the IMGXF_JPEG_E_* code of file i; the call returns IMGXF_OK unless a capacity
 * is too small (IMGXF_ERR_WORKSPACE).  Replaces Image.open's header parsing, /root/reference/transformation.py:83. */
int imgxf_jpeg_layout_host(const uint8_t* const* files, const size_t* sizes, int n, imgxf_jpeg_dec_image* images,
                           imgxf_jpeg_dec_lut* luts, int lut_cap, int* n_luts, uint16_t* quants, int quant_cap, int* n_quants,
                           uint8_t* scan, size_t scan_cap, size_t* scan_bytes, int64_t* seg_off, int32_t* seg_len, int seg_cap,
                           int* n_segs, int64_t* coef_total, int64_t* plane_total, int32_t* status);
/* Dequantisation + jpeg_idct_islow of every block of n images into their sample planes. */
int imgxf_jpeg_decode_idct(const int16_t* coefs, const imgxf_jpeg_dec_image* images, const imgxf_jpeg_dec_image* images_host,
                           int n, const uint16_t* quants, uint8_t* planes, void* stream);
/* Fancy upsampling + YCbCr -> RGB (or gray -> RGB) of n images into out[] (HWC uint8 RGB at out_off / out_pitch). */
int imgxf_jpeg_decode_color(const uint8_t* planes, const imgxf_jpeg_dec_image* images, const imgxf_jpeg_dec_image* images_host,
                            int n, uint8_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IMGXF_H */
