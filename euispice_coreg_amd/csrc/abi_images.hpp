// Part of libcoreg_hip.so's ONE translation unit (coreg_hip.hip includes the parts in order; round 6 split by concern,
// no behaviour change): C ABI: image to align, thresholds, reference on the grid / prepared on the GPU, resampling.
#pragma once
extern "C" {
int coreg_set_small(coreg_handle* h, const double* img, int32_t ny, int32_t nx) {
    if (!h) return COREG_EINVAL;
    if (!img || ny < 1 || nx < 1 || too_many(ny, nx)) return fail(h, COREG_EINVAL, "set_small: bad image");
    RETCHK(bind_device(h));
    const size_t n = (size_t)ny * nx;
    RETCHK(upload_image(h, img, n, h->small, &h->small_f32));
    h->sW = nx;
    h->sH = ny;
    if (h->small_f32)
        RETCHK(device_mean<float>(h, h->small.as<float>(), (long long)n, h->pivots.as<double>() + 1));
    else
        RETCHK(device_mean<double>(h, h->small.as<double>(), (long long)n, h->pivots.as<double>() + 1));
    return COREG_OK;
}

int coreg_set_small_f32(coreg_handle* h, const float* img, int32_t ny, int32_t nx) {
    if (!h) return COREG_EINVAL;
    if (!img || ny < 1 || nx < 1 || too_many(ny, nx)) return fail(h, COREG_EINVAL, "set_small_f32: bad image");
    trace("set_small_f32: enter");
    RETCHK(bind_device(h));
    const size_t n = (size_t)ny * nx;
    HIPCHK(h->small.reserve(n * sizeof(float)));
    // through pinned staging: the caller's buffer is free again on return, the copy itself is asynchronous -- and on the
    // upload stream, so that a reference preparation called next does not wait for it
    hipStream_t s;
    RETCHK(begin_small_upload(h, &s));
    if (h->opt_async_upload && s != h->stream) {
        h->small_f32 = true;
        h->sW = nx;
        h->sH = ny;
        void* dev = h->small.p;
        post_upload(h, [h, dev, img, n, s] { return upload_small_worker(h, dev, img, n, false, s); });
        h->small_pending = true;  // (join_small: waits for the worker to have issued everything, then for ev_small)
        trace("set_small_f32: handed to the upload thread");
        return COREG_OK;
    }
    RETCHK(staged_upload(h, h->small.p, img, n * sizeof(float), s));
    h->small_f32 = true;
    h->sW = nx;
    h->sH = ny;
    RETCHK(device_mean<float>(h, h->small.as<float>(), (long long)n, h->pivots.as<double>() + 1, s));
    trace("set_small_f32: issued");
    return end_small_upload(h, s);
}

// image to align from pinned host memory or from this GPU's memory (one asynchronous copy, no staging)
static int set_small_direct(coreg_handle* h, const void* img, int dtype, int32_t ny, int32_t nx, SrcKind kind) {
    if (!h) return COREG_EINVAL;
    if (!img || ny < 1 || nx < 1 || too_many(ny, nx) || (dtype != COREG_F32 && dtype != COREG_F64))
        return fail(h, COREG_EINVAL, "set_small: bad argument");
    RETCHK(bind_device(h));
    const size_t n = (size_t)ny * nx;
    if (dtype == COREG_F32) {
        HIPCHK(h->small.reserve(n * sizeof(float)));
        HIPCHK(hipMemcpyAsync(h->small.p, img, n * sizeof(float),
                              kind == SRC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
        h->small_f32 = true;
    } else {
        RETCHK(upload_image(h, (const double*)img, n, h->small, &h->small_f32, kind));
    }
    h->sW = nx;
    h->sH = ny;
    if (h->small_f32) return device_mean<float>(h, h->small.as<float>(), (long long)n, h->pivots.as<double>() + 1);
    return device_mean<double>(h, h->small.as<double>(), (long long)n, h->pivots.as<double>() + 1);
}

int coreg_set_small_from_device(coreg_handle* h, const void* dev_img, int dtype, int32_t ny, int32_t nx) {
    return set_small_direct(h, dev_img, dtype, ny, nx, SRC_DEVICE);
}

// image to align as the FITS data unit stores it (host or page-locked memory): raw bytes up, decode on the GPU
static int set_small_fits(coreg_handle* h, const coreg_fits_pixels* px, int32_t ny, int32_t nx, SrcKind kind) {
    if (!h) return COREG_EINVAL;
    PixFmt fmt;
    RETCHK(check_fits(h, px, &fmt));
    if (ny < 1 || nx < 1 || too_many(ny, nx)) return fail(h, COREG_EINVAL, "set_small_fits: bad image size");
    RETCHK(bind_device(h));
    const size_t n = (size_t)ny * nx, eb = fmt.elem();
    DevBuf& dst = fmt.swap_only() ? h->small : h->up_raw;
    HIPCHK(dst.reserve(n * eb));
    if (fmt.swap_only() && kind == SRC_HOST) {
        // BITPIX = -32 from host memory (what an EUI level-2 file without tile compression holds): upload stream
        hipStream_t s;
        RETCHK(begin_small_upload(h, &s));
        if (h->opt_async_upload && s != h->stream) {
            h->small_f32 = true;
            h->sW = nx;
            h->sH = ny;
            void* dev = dst.p;
            const void* src = px->data;
            post_upload(h, [h, dev, src, n, s] { return upload_small_worker(h, dev, src, n, true, s); });
            h->small_pending = true;
            return COREG_OK;
        }
        RETCHK(staged_upload(h, dst.p, px->data, n * eb, s));
        const int nb = (int)std::min<size_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(k_fits_swap32, dim3(nb), dim3(256), 0, s, (unsigned int*)dst.p, (long long)n);
        HIPCHK(hipGetLastError());
        h->small_f32 = true;
        h->sW = nx;
        h->sH = ny;
        RETCHK(device_mean<float>(h, h->small.as<float>(), (long long)n, h->pivots.as<double>() + 1, s));
        return end_small_upload(h, s);
    }
    if (kind == SRC_PINNED) HIPCHK(hipMemcpyAsync(dst.p, px->data, n * eb, hipMemcpyHostToDevice, h->stream));
    else if (kind == SRC_DEVICE) HIPCHK(hipMemcpyAsync(dst.p, px->data, n * eb, hipMemcpyDeviceToDevice, h->stream));
    else RETCHK(staged_upload(h, dst.p, px->data, n * eb));
    RETCHK(fits_decode(h, fmt, dst.p, n, h->small, &h->small_f32));
    h->sW = nx;
    h->sH = ny;
    if (h->small_f32) return device_mean<float>(h, h->small.as<float>(), (long long)n, h->pivots.as<double>() + 1);
    return device_mean<double>(h, h->small.as<double>(), (long long)n, h->pivots.as<double>() + 1);
}

int coreg_set_small_fits(coreg_handle* h, const coreg_fits_pixels* px, int32_t ny, int32_t nx) {
    return set_small_fits(h, px, ny, nx, SRC_HOST);
}

int coreg_set_small_tiled(coreg_handle* h, const coreg_fits_tiled* t) {
    if (!h) return COREG_EINVAL;
    RETCHK(bind_device(h));
    RETCHK(decode_tiled_device(h, t, h->small, &h->small_f32));
    h->sW = t->naxis1;
    h->sH = t->naxis2;
    const long long n = (long long)h->sW * h->sH;
    if (h->small_f32) return device_mean<float>(h, h->small.as<float>(), n, h->pivots.as<double>() + 1);
    return device_mean<double>(h, h->small.as<double>(), n, h->pivots.as<double>() + 1);
}

int coreg_decode_tiled_host(const coreg_fits_tiled* t, void* out, int dtype, int32_t* tile_status) {
    if (check_tiled(t) || !out || (dtype != COREG_F32 && dtype != COREG_F64)) return COREG_EINVAL;
    if (dtype == COREG_F32 && t->zbitpix != -32) return COREG_EINVAL;
    static const std::vector<float> randoms = [] {
        std::vector<float> r(coregrice::kNRandom);
        coregrice::init_randoms(r.data());
        return r;
    }();
    coregrice::TileImage im;
    fill_tile_image(*t, &im);
    im.heap = (const unsigned char*)t->heap;
    im.tile_offset = t->tile_offset;
    im.tile_nbytes = t->tile_nbytes;
    im.zscale = t->zscale;
    im.zzero = t->zzero;
    im.randoms = randoms.data();
    im.out = out;
    im.out_dtype = dtype == COREG_F32 ? coregrice::OUT_F32 : coregrice::OUT_F64;
    const int nt = t->n_tiles;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int n_thr = (int)std::min<long long>(std::min<unsigned>(hw, 12u),
                                               std::max<long long>(1, (long long)t->naxis1 * t->naxis2 / (1 << 16)));
    auto work = [&](int lo, int hi) {
        for (int k = lo; k < hi; ++k) {
            const int e = coregrice::decode_tile(im, k);
            if (tile_status) tile_status[k] = e;
        }
    };
    if (n_thr <= 1) {
        work(0, nt);
        return COREG_OK;
    }
    std::vector<std::thread> th;
    const int per = (nt + n_thr - 1) / n_thr;
    for (int k = 1; k < n_thr; ++k) th.emplace_back(work, std::min(nt, k * per), std::min(nt, (k + 1) * per));
    work(0, std::min(nt, per));
    for (auto& x : th) x.join();
    return COREG_OK;
}

int coreg_encode_tiled_host(const void* pixels, int dtype, int ny, int nx, int tile_x, int tile_y, int bytepix,
                            int blocksize, int quantize, int dither0, double scale, unsigned char* heap,
                            long long heap_cap, int32_t* tile_nbytes, int64_t* tile_offset, double* zscale, double* zzero,
                            long long* heap_used) {
    if (!pixels || !heap || !tile_nbytes || !tile_offset || !heap_used || ny <= 0 || nx <= 0 || tile_x <= 0 || tile_y <= 0)
        return COREG_EINVAL;
    if (blocksize <= 0 || blocksize > 1024 || (bytepix != 1 && bytepix != 2 && bytepix != 4)) return COREG_EINVAL;
    const bool is_float = dtype == COREG_F32 || dtype == COREG_F64;
    if (is_float) {
        if (quantize < coregrice::Q_NO_DITHER || quantize > coregrice::Q_DITHER_2 || !(scale > 0) || !std::isfinite(scale) ||
            !zscale || !zzero || bytepix != 4)
            return COREG_EINVAL;
    } else if (dtype != COREG_I32) {
        return COREG_EINVAL;  // integer images: the stored integers as int32, whatever BYTEPIX
    }
    static const std::vector<float> randoms = [] {
        std::vector<float> r(coregrice::kNRandom);
        coregrice::init_randoms(r.data());
        return r;
    }();
    coregrice::TileImage t{};
    t.naxis1 = nx;
    t.naxis2 = ny;
    t.ztile1 = tile_x;
    t.ztile2 = tile_y;
    t.dither0 = dither0;
    const int ntx = (nx + tile_x - 1) / tile_x, nty = (ny + tile_y - 1) / tile_y;
    std::vector<int32_t> q((size_t)tile_x * tile_y);
    long long used = 0;
    for (int n = 0; n < ntx * nty; ++n) {
        const coregrice::TileBox b = coregrice::tile_box(t, n);
        const int npx = b.tw * b.th;
        if (is_float) {
            const int iseed = coregrice::dither_seed(t, n);
            const int e = dtype == COREG_F32
                              ? coregrice::quantize_tile((const float*)pixels, nx, b, quantize, iseed, randoms.data(), scale,
                                                         q.data(), &zzero[n])
                              : coregrice::quantize_tile((const double*)pixels, nx, b, quantize, iseed, randoms.data(), scale,
                                                         q.data(), &zzero[n]);
            if (e) return COREG_EINVAL;  // (the tile's range does not fit 32-bit integers at this scale)
            zscale[n] = scale;
        } else {
            const int32_t* src = (const int32_t*)pixels;
            for (int y = 0; y < b.th; ++y)
                std::memcpy(q.data() + (size_t)y * b.tw, src + (size_t)(b.y0 + y) * nx + b.x0, (size_t)b.tw * 4);
        }
        const int64_t len = coregrice::rice_encode_tile(q.data(), npx, blocksize, bytepix, heap + used, heap_cap - used);
        if (len < 0) return COREG_ENOMEM;
        tile_offset[n] = used;
        tile_nbytes[n] = (int32_t)len;
        used += len;
    }
    *heap_used = used;
    return COREG_OK;
}

int coreg_threshold_small(coreg_handle* h, int has_min, double vmin, int has_max, double vmax, long long* n_finite) {
    if (!h) return COREG_EINVAL;
    if (!h->small.p) return fail(h, COREG_ESTATE, "coreg_set_small has not been called");
    RETCHK(bind_device(h));
    const long long n = (long long)h->sW * h->sH;
    if (has_min || has_max) {
        const int nb = (int)std::min<long long>((n + 255) / 256, 2048);
        if (h->small_f32)
            hipLaunchKernelGGL((k_threshold<float>), dim3(nb), dim3(256), 0, h->stream, h->small.as<float>(), n, has_min,
                               vmin, has_max, vmax);
        else
            hipLaunchKernelGGL((k_threshold<double>), dim3(nb), dim3(256), 0, h->stream, h->small.as<double>(), n, has_min,
                               vmin, has_max, vmax);
        HIPCHK(hipGetLastError());
    }
    // pivot = mean of what is left (same value as uploading a host-thresholded image)
    if (h->small_f32)
        RETCHK(device_mean<float>(h, h->small.as<float>(), n, h->pivots.as<double>() + 1));
    else
        RETCHK(device_mean<double>(h, h->small.as<double>(), n, h->pivots.as<double>() + 1));
    if (n_finite) {
        long long cnt[256];
        HIPCHK(hipMemcpyAsync(cnt, h->red_cnt.p, sizeof(cnt), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        long long c = 0;
        for (int i = 0; i < 256; ++i) c += cnt[i];
        *n_finite = c;
    }
    return COREG_OK;
}

static int ref_pivot(coreg_handle* h) {
    const long long n = (long long)h->gW * h->gH;
    if (h->ref_dtype == COREG_F32) return device_mean<float>(h, h->ref.as<float>(), n, h->pivots.as<double>());
    return device_mean<double>(h, h->ref.as<double>(), n, h->pivots.as<double>());
}

int coreg_set_reference_on_grid(coreg_handle* h, const void* ref, int dtype, int32_t gy, int32_t gx) {
    if (!h) return COREG_EINVAL;
    if (!ref || gy < 1 || gx < 1 || (dtype != COREG_F32 && dtype != COREG_F64))
        return fail(h, COREG_EINVAL, "set_reference_on_grid: bad argument");
    RETCHK(bind_device(h));
    const size_t bytes = (size_t)gy * gx * (dtype == COREG_F32 ? 4 : 8);
    HIPCHK(h->ref.reserve(bytes));
    HIPCHK(hipMemcpyAsync(h->ref.p, ref, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->gW = gx;
    h->gH = gy;
    h->ref_dtype = dtype;
    return ref_pivot(h);
}

// Which pixels of the W x H source image can the once-only resample touch?  The bounding box of the in-bounds sample
// coordinates is computed on the GPU by the same coordinate function the resample uses (k_resample_bbox), widened by the
// spline apron and clipped to the image.  Uploading only that rectangle -- the Carrington grid of the headline touches
// 2 % of the 3072 x 3072 reference, the sub-map of a 2048 x 2048 HRIEUV field 0.6 % -- takes the reference image out of
// the PCIe-inclusive cost of a call; results are bit-identical (same pixels, same arithmetic).  Costs one ~20 us kernel
// and a 4-double read-back.  crop = {0, 0, W, H} when cropping would not pay (more than half the image) or is disabled
// (coreg_set_option "crop_reference" 0).
struct CropRect {
    int x0, y0, w, h;
};
static int reference_crop(coreg_handle* h, int mode, const ResampleArgs& a0, int order, CropRect* out) {
    *out = {0, 0, a0.W, a0.H};
    if (!h->opt_crop_reference || a0.W < 64 || a0.H < 64) return COREG_OK;
    const int nb = 256;
    // On a side stream: the box depends on headers and grid tables only, so it need not queue behind the upload of the
    // image to align that usually precedes it on the handle's stream (its last DMA segment is still in flight).
    if (!h->aux_stream) HIPCHK(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
    HIPCHK(h->bbox_buf.reserve((size_t)nb * 4 * sizeof(double)));
    ResampleArgs a = a0;
    a.bbox = h->bbox_buf.as<double>();
    if (mode == MODE_TRANSLATE)
        hipLaunchKernelGGL((k_resample_bbox<MODE_TRANSLATE>), dim3(nb), dim3(256), 0, h->aux_stream, a);
    else if (mode == MODE_CAR)
        hipLaunchKernelGGL((k_resample_bbox<MODE_CAR>), dim3(nb), dim3(256), 0, h->aux_stream, a);
    else
        hipLaunchKernelGGL((k_resample_bbox<MODE_HOMOGRAPHY>), dim3(nb), dim3(256), 0, h->aux_stream, a);
    HIPCHK(hipGetLastError());
    std::vector<double> part((size_t)nb * 4);
    HIPCHK(hipMemcpyAsync(part.data(), a.bbox, part.size() * sizeof(double), hipMemcpyDeviceToHost, h->aux_stream));
    HIPCHK(hipStreamSynchronize(h->aux_stream));
    double mnx = 1e300, mxx = -1e300, mny = 1e300, mxy = -1e300;
    for (int b = 0; b < nb; ++b) {
        mnx = std::min(mnx, part[4 * b + 0]);
        mxx = std::max(mxx, part[4 * b + 1]);
        mny = std::min(mny, part[4 * b + 2]);
        mxy = std::max(mxy, part[4 * b + 3]);
    }
    const int apron = order / 2 + 3;  // taps of an in-bounds sample: [floor(c) - order/2 - 1, floor(c) + order - order/2 + 1]
    int x0 = 0, x1 = apron * 2, y0 = 0, y1 = apron * 2;  // nothing in bounds: any small rectangle (never read)
    if (mnx <= mxx && mny <= mxy) {
        x0 = std::max(0, (int)std::floor(mnx) - apron);
        x1 = std::min(a0.W - 1, (int)std::floor(mxx) + apron + 1);
        y0 = std::max(0, (int)std::floor(mny) - apron);
        y1 = std::min(a0.H - 1, (int)std::floor(mxy) + apron + 1);
    }
    // (taps that mirror at an image edge stay inside: the rectangle starts AT that edge and is at least 2 aprons wide)
    x1 = std::min(a0.W - 1, std::max(x1, x0 + 2 * apron));
    y1 = std::min(a0.H - 1, std::max(y1, y0 + 2 * apron));
    const long long area = (long long)(x1 - x0 + 1) * (y1 - y0 + 1);
    if (2 * area > (long long)a0.W * a0.H) return COREG_OK;  // not worth a strided copy
    *out = {x0, y0, x1 - x0 + 1, y1 - y0 + 1};
    return COREG_OK;
}

// rows y0 .. of a host image, columns x0 .., packed into pinned staging and sent to `dev` (contiguous, pitch = crop width)
static int staged_upload_rect(coreg_handle* h, void* dev, const void* host, size_t elem, int W, const CropRect& c) {
    const char* src = (const char*)host + ((size_t)c.y0 * W + c.x0) * elem;
    if (c.w == W) return staged_upload(h, dev, src, (size_t)c.w * c.h * elem);  // whole rows: one contiguous range
    const int k = h->pin_img_next;
    h->pin_img_next ^= 1;
    if (!h->ev_img[k]) HIPCHK(hipEventCreateWithFlags(&h->ev_img[k], hipEventDisableTiming));
    else HIPCHK(hipEventSynchronize(h->ev_img[k]));
    const size_t bytes = (size_t)c.w * c.h * elem;
    HIPCHK(h->pin_img[k].reserve(bytes));
    parallel_copy_rows(h->pin_img[k].p, src, (size_t)c.h, (size_t)c.w * elem, (size_t)W * elem);
    HIPCHK(hipMemcpyAsync(dev, h->pin_img[k].p, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipEventRecord(h->ev_img[k], h->stream));
    return COREG_OK;
}

// the reference image's pixels: float64 from the caller (tested for float32-exactness on the GPU), the float32
// pixels a BITPIX=-32 FITS file holds (half the PCIe bytes; the reference's float64 cast of them is exact), or the raw
// big-endian elements of the FITS data unit (decoded on the GPU)
// (src_on_device: the pixels are read where they are, by the resample kernel on the handle's stream -- no copy)
static int upload_reference_source(coreg_handle* h, const void* large, size_t n, const PixFmt& fmt, bool* f32,
                                   SrcKind kind, const void** img_dev, int W = 0, const CropRect* crop = nullptr) {
    const bool src_f32 = fmt.f32;
    if (kind == SRC_DEVICE) {
        if (fmt.raw()) return fail(h, COREG_ENOTIMPL, "raw FITS pixels must come from host memory");
        *f32 = src_f32;
        *img_dev = large;
        return COREG_OK;
    }
    const bool cropped = crop && kind == SRC_HOST && (size_t)crop->w * crop->h < n;
    const size_t nc = cropped ? (size_t)crop->w * crop->h : n;
    if (fmt.raw()) {
        // only the rectangle the resample can touch crosses PCIe, as stored in the file; decoded on the device
        const size_t eb = fmt.elem();
        DevBuf& dst = fmt.swap_only() ? h->tmp_img : h->up_raw;
        HIPCHK(dst.reserve(nc * eb));
        if (cropped) RETCHK(staged_upload_rect(h, dst.p, large, eb, W, *crop));
        else if (kind == SRC_PINNED) HIPCHK(hipMemcpyAsync(dst.p, large, n * eb, hipMemcpyHostToDevice, h->stream));
        else RETCHK(staged_upload(h, dst.p, large, n * eb));
        RETCHK(fits_decode(h, fmt, dst.p, nc, h->tmp_img, f32));
        *img_dev = h->tmp_img.p;
        return COREG_OK;
    }
    if (cropped) {
        // only the rectangle the resample can touch crosses PCIe
        if (src_f32) {
            HIPCHK(h->tmp_img.reserve(nc * sizeof(float)));
            RETCHK(staged_upload_rect(h, h->tmp_img.p, large, sizeof(float), W, *crop));
            *f32 = true;
        } else {
            HIPCHK(h->up_f64.reserve(nc * sizeof(double)));
            RETCHK(staged_upload_rect(h, h->up_f64.p, large, sizeof(double), W, *crop));
            RETCHK(upload_image(h, h->up_f64.as<double>(), nc, h->tmp_img, f32, SRC_DEVICE));
        }
        *img_dev = h->tmp_img.p;
        return COREG_OK;
    }
    if (!src_f32) {
        RETCHK(upload_image(h, (const double*)large, n, h->tmp_img, f32, kind));
    } else {
        HIPCHK(h->tmp_img.reserve(n * sizeof(float)));
        if (kind == SRC_PINNED)
            HIPCHK(hipMemcpyAsync(h->tmp_img.p, large, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
        else
            RETCHK(staged_upload(h, h->tmp_img.p, large, n * sizeof(float)));
        *f32 = true;
    }
    *img_dev = h->tmp_img.p;
    return COREG_OK;
}

static int prepare_carrington(coreg_handle* h, const void* large, const PixFmt& fmt, int32_t ny, int32_t nx,
                              const coreg_wcs2d* hdr, const coreg_carr_grid* grid, double solar_r, int order,
                              SrcKind kind = SRC_HOST) {
    if (!h) return COREG_EINVAL;
    if (!large || !hdr || !grid || ny < 1 || nx < 1 || too_many(ny, nx))
        return fail(h, COREG_EINVAL, "prepare_reference: bad argument");
    RETCHK(check_order(h, order));
    RETCHK(check_wcs(h, hdr, true));
    RETCHK(check_grid(h, grid));
    if (!std::isfinite(solar_r) || !(solar_r > 0.0)) return fail(h, COREG_EINVAL, "solar_r must be positive");
    trace("prepare_carrington: enter");
    RETCHK(bind_device_nowait(h));  // (touches neither the image to align nor its pivot: no join with the upload stream)
    ResampleArgs a;
    std::memset(&a, 0, sizeof(a));
    RETCHK(upload_carr_tables(h, *grid, *hdr, &a.carr));
    set_carr_common(&a.carr, carr_common(*hdr, solar_r));
    carr_origin(*hdr, &a.x0, &a.y0);
    a.W = nx;
    a.H = ny;
    a.gw = grid->n_lon;
    a.gh = grid->n_lat;
    CropRect crop = {0, 0, nx, ny};
    if (kind == SRC_HOST) RETCHK(reference_crop(h, MODE_TRANSLATE, a, order, &crop));
    trace("prepare_carrington: crop box known");
    bool f32;
    const void* img_dev = nullptr;
    RETCHK(upload_reference_source(h, large, (size_t)ny * nx, fmt, &f32, kind, &img_dev, nx, &crop));
    a.img = img_dev;
    if (crop.w != nx || crop.h != ny) a.crop = {crop.x0, crop.y0, crop.w};
    HIPCHK(h->ref.reserve((size_t)a.gw * a.gh * sizeof(double)));
    a.out = h->ref.p;
    RETCHK(dispatch_resample(h, MODE_TRANSLATE, order, f32, false, a, 0));
    h->gW = a.gw;
    h->gH = a.gh;
    h->ref_dtype = COREG_F64;
    RETCHK(ref_pivot(h));
    trace("prepare_carrington: issued");
    // no host sync: the pinned staging is guarded by staged_upload's own wait, everything else is stream-ordered
    return COREG_OK;  // tmp_img stays allocated: the next preparation re-uses it (hipFree would stall the device)
}

int coreg_prepare_reference_carrington(coreg_handle* h, const double* large, int32_t ny, int32_t nx,
                                       const coreg_wcs2d* hdr, const coreg_carr_grid* grid, double solar_r, int order) {
    return prepare_carrington(h, large, PixFmt::native(false), ny, nx, hdr, grid, solar_r, order);
}

int coreg_prepare_reference_carrington_f32(coreg_handle* h, const float* large, int32_t ny, int32_t nx,
                                           const coreg_wcs2d* hdr, const coreg_carr_grid* grid, double solar_r,
                                           int order) {
    return prepare_carrington(h, large, PixFmt::native(true), ny, nx, hdr, grid, solar_r, order);
}

static int prepare_helioprojective(coreg_handle* h, const void* large, const PixFmt& fmt, int32_t ny, int32_t nx,
                                   const coreg_wcs2d* hdr_large, const coreg_wcs2d* hdr_small, int order,
                                   SrcKind kind = SRC_HOST) {
    if (!h) return COREG_EINVAL;
    if (!large || !hdr_large || !hdr_small || ny < 1 || nx < 1 || too_many(ny, nx))
        return fail(h, COREG_EINVAL, "prepare_reference: bad argument");
    if (hdr_small->naxis1 < 1 || hdr_small->naxis2 < 1 || too_many(hdr_small->naxis1, hdr_small->naxis2))
        return fail(h, COREG_EINVAL, "hdr_small: NAXIS1/2 missing (or more than 2^31 - 1 pixels)");
    if (hdr_large->proj != hdr_small->proj || (hdr_small->proj != COREG_PROJ_TAN && hdr_small->proj != COREG_PROJ_CAR))
        return fail(h, COREG_ENOTIMPL, "prepare_reference_helioprojective: both headers TAN, or both CAR");
    RETCHK(check_order(h, order));
    RETCHK(check_wcs(h, hdr_large, false));
    RETCHK(check_wcs(h, hdr_small, false));
    RETCHK(bind_device_nowait(h));
    ResampleArgs a;
    std::memset(&a, 0, sizeof(a));
    int mode = MODE_HOMOGRAPHY;
    if (hdr_small->proj == COREG_PROJ_CAR) {
        // two Carrington maps (align_using_initial_carrington: both branches of alignment.py:649-651 / :765-767 build the
        // sub-map for this frame too): pixel of hdr_small -> native angles -> sphere rotation -> native angles of hdr_large
        // -> its pixel, the per-lag map of sweep_car with the roles of the two maps exchanged
        mode = MODE_CAR;
        Mat3 r_small, r_large;
        if (car_native_to_celestial(*hdr_small, &r_small) || car_native_to_celestial(*hdr_large, &r_large))
            return fail(h, COREG_EINVAL, "prepare_reference: no valid native pole for this CRVAL2 / LONPOLE (CAR)");
        const Mat3 m = mat_mul(mat_T(r_large), r_small);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) a.hom.h[3 * i + j] = (double)m.m[i][j];
        const Affine2 fwd = car_pix_to_native(*hdr_small), inv = car_native_to_pix(*hdr_large);
        a.car_fwd.m00 = fwd.m00;
        a.car_fwd.m01 = fwd.m01;
        a.car_fwd.m10 = fwd.m10;
        a.car_fwd.m11 = fwd.m11;
        a.car_fwd.b0 = fwd.b0;
        a.car_fwd.b1 = fwd.b1;
        a.car_inv.m00 = inv.m00;
        a.car_inv.m01 = inv.m01;
        a.car_inv.m10 = inv.m10;
        a.car_inv.m11 = inv.m11;
        a.car_inv.b0 = inv.b0;
        a.car_inv.b1 = inv.b1;
    } else {
        homography(*hdr_small, *hdr_large, a.hom.h);  // alignment.py:993: pixels of hdr_cut -> pixels of hdr_large
    }
    a.W = nx;
    a.H = ny;
    a.gw = hdr_small->naxis1;
    a.gh = hdr_small->naxis2;
    CropRect crop = {0, 0, nx, ny};
    if (kind == SRC_HOST) RETCHK(reference_crop(h, mode, a, order, &crop));
    bool f32;
    const void* img_dev = nullptr;
    RETCHK(upload_reference_source(h, large, (size_t)ny * nx, fmt, &f32, kind, &img_dev, nx, &crop));
    a.img = img_dev;
    if (crop.w != nx || crop.h != ny) a.crop = {crop.x0, crop.y0, crop.w};
    HIPCHK(h->ref.reserve((size_t)a.gw * a.gh * sizeof(float)));
    a.out = h->ref.p;
    RETCHK(dispatch_resample(h, mode, order, f32, true, a, 0));
    h->gW = a.gw;
    h->gH = a.gh;
    h->ref_dtype = COREG_F32;
    RETCHK(ref_pivot(h));
    return COREG_OK;
}

int coreg_prepare_reference_helioprojective(coreg_handle* h, const double* large, int32_t ny, int32_t nx,
                                            const coreg_wcs2d* hdr_large, const coreg_wcs2d* hdr_small, int order) {
    return prepare_helioprojective(h, large, PixFmt::native(false), ny, nx, hdr_large, hdr_small, order);
}

int coreg_prepare_reference_helioprojective_f32(coreg_handle* h, const float* large, int32_t ny, int32_t nx,
                                                const coreg_wcs2d* hdr_large, const coreg_wcs2d* hdr_small,
                                                int order) {
    return prepare_helioprojective(h, large, PixFmt::native(true), ny, nx, hdr_large, hdr_small, order);
}

int coreg_prepare_reference_carrington_from_device(coreg_handle* h, const void* dev_large, int dtype, int32_t ny,
                                                   int32_t nx, const coreg_wcs2d* hdr_large,
                                                   const coreg_carr_grid* grid, double solar_r, int order) {
    if (h && dtype != COREG_F32 && dtype != COREG_F64) return fail(h, COREG_EINVAL, "prepare_reference: bad dtype");
    return prepare_carrington(h, dev_large, PixFmt::native(dtype == COREG_F32), ny, nx, hdr_large, grid, solar_r, order,
                              SRC_DEVICE);
}

int coreg_prepare_reference_helioprojective_from_device(coreg_handle* h, const void* dev_large, int dtype, int32_t ny,
                                                        int32_t nx, const coreg_wcs2d* hdr_large,
                                                        const coreg_wcs2d* hdr_small, int order) {
    if (h && dtype != COREG_F32 && dtype != COREG_F64) return fail(h, COREG_EINVAL, "prepare_reference: bad dtype");
    return prepare_helioprojective(h, dev_large, PixFmt::native(dtype == COREG_F32), ny, nx, hdr_large, hdr_small, order,
                                   SRC_DEVICE);
}

int coreg_prepare_reference_carrington_fits(coreg_handle* h, const coreg_fits_pixels* px, int32_t ny, int32_t nx,
                                            const coreg_wcs2d* hdr_large, const coreg_carr_grid* grid, double solar_r,
                                            int order) {
    if (!h) return COREG_EINVAL;
    PixFmt fmt;
    RETCHK(check_fits(h, px, &fmt));
    return prepare_carrington(h, px->data, fmt, ny, nx, hdr_large, grid, solar_r, order);
}

int coreg_prepare_reference_helioprojective_fits(coreg_handle* h, const coreg_fits_pixels* px, int32_t ny, int32_t nx,
                                                 const coreg_wcs2d* hdr_large, const coreg_wcs2d* hdr_small, int order) {
    if (!h) return COREG_EINVAL;
    PixFmt fmt;
    RETCHK(check_fits(h, px, &fmt));
    return prepare_helioprojective(h, px->data, fmt, ny, nx, hdr_large, hdr_small, order);
}

int coreg_prepare_reference_carrington_tiled(coreg_handle* h, const coreg_fits_tiled* t, const coreg_wcs2d* hdr_large,
                                             const coreg_carr_grid* grid, double solar_r, int order) {
    if (!h) return COREG_EINVAL;
    RETCHK(bind_device(h));
    bool f32;
    RETCHK(decode_tiled_device(h, t, h->dec_img, &f32));  // the compressed bytes cross PCIe, the pixels never do
    return prepare_carrington(h, h->dec_img.p, PixFmt::native(f32), t->naxis2, t->naxis1, hdr_large, grid, solar_r, order,
                              SRC_DEVICE);
}

int coreg_prepare_reference_helioprojective_tiled(coreg_handle* h, const coreg_fits_tiled* t, const coreg_wcs2d* hdr_large,
                                                  const coreg_wcs2d* hdr_small, int order) {
    if (!h) return COREG_EINVAL;
    RETCHK(bind_device(h));
    bool f32;
    RETCHK(decode_tiled_device(h, t, h->dec_img, &f32));
    return prepare_helioprojective(h, h->dec_img.p, PixFmt::native(f32), t->naxis2, t->naxis1, hdr_large, hdr_small, order,
                                   SRC_DEVICE);
}

int coreg_get_reference_on_grid(coreg_handle* h, void* out, int dtype) {
    if (!h) return COREG_EINVAL;
    if (!h->ref.p) return fail(h, COREG_ESTATE, "no reference image on the target grid");
    if (!out || dtype != h->ref_dtype) return fail(h, COREG_EINVAL, "get_reference_on_grid: dtype mismatch");
    RETCHK(bind_device(h));
    const size_t bytes = (size_t)h->gW * h->gH * (dtype == COREG_F32 ? 4 : 8);
    HIPCHK(hipMemcpyAsync(out, h->ref.p, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return COREG_OK;
}

int coreg_resample_carrington(coreg_handle* h, const coreg_wcs2d* hdr, const coreg_carr_grid* grid, double solar_r,
                              int order, double* out) {
    if (!h) return COREG_EINVAL;
    if (!hdr || !grid || !out) return fail(h, COREG_EINVAL, "resample_carrington: bad argument");
    if (!h->small.p) return fail(h, COREG_ESTATE, "coreg_set_small has not been called");
    RETCHK(check_order(h, order));
    RETCHK(check_wcs(h, hdr, true));
    RETCHK(check_grid(h, grid));
    if (!std::isfinite(solar_r) || !(solar_r > 0.0)) return fail(h, COREG_EINVAL, "solar_r must be positive");
    RETCHK(bind_device(h));
    ResampleArgs a;
    std::memset(&a, 0, sizeof(a));
    RETCHK(upload_carr_tables(h, *grid, *hdr, &a.carr));
    set_carr_common(&a.carr, carr_common(*hdr, solar_r));
    carr_origin(*hdr, &a.x0, &a.y0);
    a.img = h->small.p;
    a.W = h->sW;
    a.H = h->sH;
    a.gw = grid->n_lon;
    a.gh = grid->n_lat;
    const size_t bytes = (size_t)a.gw * a.gh * sizeof(double);
    HIPCHK(h->out_dev.reserve(bytes));
    a.out = h->out_dev.p;
    RETCHK(dispatch_resample(h, MODE_TRANSLATE, order, h->small_f32, false, a, 0));
    HIPCHK(hipMemcpyAsync(out, h->out_dev.p, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return COREG_OK;
}

static int resample_helio(coreg_handle* h, const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr, int order, void* out,
                          bool out_f32) {
    if (!h) return COREG_EINVAL;
    if (!hdr_target || !hdr || !out) return fail(h, COREG_EINVAL, "resample_helioprojective: bad argument");
    if (!h->small.p) return fail(h, COREG_ESTATE, "coreg_set_small has not been called");
    if (hdr_target->naxis1 < 1 || hdr_target->naxis2 < 1 || too_many(hdr_target->naxis1, hdr_target->naxis2))
        return fail(h, COREG_EINVAL, "hdr_target: NAXIS missing (or more than 2^31 - 1 pixels)");
    RETCHK(check_order(h, order));
    RETCHK(check_wcs(h, hdr_target, false));
    RETCHK(check_wcs(h, hdr, false));
    RETCHK(bind_device(h));
    ResampleArgs a;
    std::memset(&a, 0, sizeof(a));
    if (hdr_target->proj != COREG_PROJ_TAN || hdr->proj != COREG_PROJ_TAN)
        return fail(h, COREG_ENOTIMPL, "resample_helioprojective: TAN headers only");
    homography(*hdr_target, *hdr, a.hom.h);  // alignment.py:1022
    a.img = h->small.p;
    a.W = h->sW;
    a.H = h->sH;
    a.gw = hdr_target->naxis1;
    a.gh = hdr_target->naxis2;
    const size_t bytes = (size_t)a.gw * a.gh * (out_f32 ? sizeof(float) : sizeof(double));
    HIPCHK(h->out_dev.reserve(bytes));
    a.out = h->out_dev.p;
    RETCHK(dispatch_resample(h, MODE_HOMOGRAPHY, order, h->small_f32, out_f32, a, 0));
    HIPCHK(hipMemcpyAsync(out, h->out_dev.p, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return COREG_OK;
}

int coreg_resample_helioprojective(coreg_handle* h, const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr, int order,
                                   float* out) {
    return resample_helio(h, hdr_target, hdr, order, out, true);
}

int coreg_resample_helioprojective_f64(coreg_handle* h, const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr,
                                       int order, double* out) {
    return resample_helio(h, hdr_target, hdr, order, out, false);
}
}  // extern "C"
