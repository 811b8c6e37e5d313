// svt_entry_seams.h -- part of the single translation unit svtyper_hip.hip (included there, in order; not a stand-alone header):
// C ABI: the array forms of the reference's inner seams (svt_bayes_gt, svt_genotype_counts).

static int svt_bayes_gt_impl(const int32_t* ref, const int32_t* alt, const uint8_t* is_dup, uint64_t n, double* out,
                 int device)
{
    if (n == 0) return SVT_OK;
    if (!ref || !alt || !is_dup || !out) return fail(SVT_ERR_INVALID, "null argument");
    const int ndev = svt_device_count();
    if (ndev <= 0) return fail(SVT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SVT_ERR_NO_DEVICE, "device index out of range");
    int64_t max_total = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (ref[i] < 0 || alt[i] < 0) return fail(SVT_ERR_INVALID, "negative read count");
        max_total = std::max<int64_t>(max_total, (int64_t)ref[i] + alt[i]);
    }
    if (max_total >= (1 << 24)) return fail(SVT_ERR_INVALID, "ref + alt must be < 2^24");
    std::vector<double> l10((size_t)max_total + 2);
    l10[0] = 0.0;
    for (size_t i = 1; i < l10.size(); ++i) l10[i] = py_log10((double)i);
    GtConsts c{};
    fill_gt_consts(c, 1.0, 1.0);
    HIP_TRY(hipSetDevice(device));
    DevScratch d_ref, d_alt, d_dup, d_l10, d_out;
    SVT_TRY(d_ref.alloc(n * sizeof(int32_t)));
    SVT_TRY(d_alt.alloc(n * sizeof(int32_t)));
    SVT_TRY(d_dup.alloc(n));
    SVT_TRY(d_l10.alloc(l10.size() * sizeof(double)));
    SVT_TRY(d_out.alloc(n * 4 * sizeof(double)));
    {
        Stager st(nullptr);
        SVT_TRY(st.copy(d_ref.p, ref, n * sizeof(int32_t)));
        SVT_TRY(st.copy(d_alt.p, alt, n * sizeof(int32_t)));
        SVT_TRY(st.copy(d_dup.p, is_dup, n));
        SVT_TRY(st.copy(d_l10.p, l10.data(), l10.size() * sizeof(double)));
        SVT_TRY(st.finish());
    }
    hipLaunchKernelGGL(svt_bayes_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0,
                       d_ref.as<int32_t>(), d_alt.as<int32_t>(), d_dup.as<uint8_t>(), n, d_l10.as<double>(), c,
                       d_out.as<double>());
    HIP_TRY(hipGetLastError());
    return d2h_staged(out, d_out.p, n * 4 * sizeof(double), nullptr);
}

int svt_bayes_gt(const int32_t* ref, const int32_t* alt, const uint8_t* is_dup, uint64_t n, double* out, int device)
{
    return guarded([&] { return svt_bayes_gt_impl(ref, alt, is_dup, n, out, device); });
}

static int svt_genotype_counts_impl(const double* counts, const uint8_t* is_dup, uint64_t n, double split_weight,
                                    double disc_weight, svt_result* out, int device)
{
    if (n == 0) return SVT_OK;
    if (!counts || !is_dup || !out) return fail(SVT_ERR_INVALID, "null argument");
    if (!(split_weight >= 0.0) || !(disc_weight >= 0.0) || !std::isfinite(split_weight) || !std::isfinite(disc_weight))
        return fail(SVT_ERR_INVALID, "weights must be finite and >= 0");
    const int ndev = svt_device_count();
    if (ndev <= 0) return fail(SVT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SVT_ERR_NO_DEVICE, "device index out of range");
    double max_total = 0.0;   // bound of QR + QA: the log10 table must reach it
    for (uint64_t i = 0; i < n; ++i) {
        const double* t = counts + 5 * i;
        for (int k = 0; k < 5; ++k)
            if (!(t[k] >= 0.0) || !std::isfinite(t[k])) return fail(SVT_ERR_INVALID, "counts must be finite and >= 0");
        max_total = std::max(max_total, split_weight * ((t[0] + t[1]) + t[2]) + disc_weight * (t[3] + t[4]));
    }
    if (max_total >= (double)(1 << 24)) return fail(SVT_ERR_INVALID, "weighted counts must stay below 2^24");
    std::vector<double> l10((size_t)max_total + 4);
    l10[0] = 0.0;
    for (size_t i = 1; i < l10.size(); ++i) l10[i] = py_log10((double)i);
    GtConsts c{};
    fill_gt_consts(c, split_weight, disc_weight);
    HIP_TRY(hipSetDevice(device));
    DevScratch d_counts, d_dup, d_l10, d_out;
    SVT_TRY(d_counts.alloc(n * 5 * sizeof(double)));
    SVT_TRY(d_dup.alloc(n));
    SVT_TRY(d_l10.alloc(l10.size() * sizeof(double)));
    SVT_TRY(d_out.alloc(n * sizeof(svt_result)));
    {
        Stager st(nullptr);
        SVT_TRY(st.copy(d_counts.p, counts, n * 5 * sizeof(double)));
        SVT_TRY(st.copy(d_dup.p, is_dup, n));
        SVT_TRY(st.copy(d_l10.p, l10.data(), l10.size() * sizeof(double)));
        SVT_TRY(st.finish());
    }
    hipLaunchKernelGGL(svt_counts_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0,
                       d_counts.as<double>(), d_dup.as<uint8_t>(), n, d_l10.as<double>(), c, d_out.as<svt_result>());
    HIP_TRY(hipGetLastError());
    return d2h_staged(out, d_out.p, n * sizeof(svt_result), nullptr);
}

int svt_genotype_counts(const double* counts, const uint8_t* is_dup, uint64_t n, double split_weight, double disc_weight,
                        svt_result* out, int device)
{
    return guarded([&] { return svt_genotype_counts_impl(counts, is_dup, n, split_weight, disc_weight, out, device); });
}

