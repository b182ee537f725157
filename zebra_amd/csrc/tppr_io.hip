// Dictionaries of the streaming T-PPR state in and out of the device rows (checkpoints, warm starts):
// zt_tppr_export / export_rows / import / import_rows.  Host-side granule (de)coding + two row gather / scatter kernels.
#include "tppr_state.hpp"

#include <vector>

using namespace zt;

namespace {
// granules of one row -> the reference's dictionary items in iteration order (zeros beyond len)
void decode_row(const u64 *r, size_t k, int32_t *len_out, double *norm_out, int64_t *eidx, int64_t *node, double *ts,
                double *w)
{
    auto lo = [](u64 x) { return (u64)(unsigned)x; };
    *len_out = (int)(unsigned)r[0];
    const u64 nb = lo(r[1]) | (lo(r[2]) << 32);
    memcpy(norm_out, &nb, 8);
    for (size_t j = 0; j < k; ++j) {
        const u64 *e = r + HDR + j;
        if ((int)j < *len_out) {
            node[j] = (int64_t)lo(e[0]);
            eidx[j] = (int64_t)lo(e[k]);
            const u64 tb = lo(e[2 * k]) | (lo(e[3 * k]) << 32), wb = lo(e[4 * k]) | (lo(e[5 * k]) << 32);
            memcpy(&ts[j], &tb, 8);
            memcpy(&w[j], &wb, 8);
        } else {
            eidx[j] = 0; node[j] = 0; ts[j] = 0.0; w[j] = 0.0;
        }
    }
}

__global__ void k_gather_rows(const u64 *__restrict__ rows, const long long *__restrict__ ids, long long n, int rg,
                              u64 *__restrict__ out)
{
    const long long r = blockIdx.x;
    if (r >= n) return;
    const u64 *src = rows + ids[r] * rg;
    for (int c = threadIdx.x; c < rg; c += blockDim.x) out[r * rg + c] = src[c];
}
}  // namespace

extern "C" int zt_tppr_export(zt_tppr *h, int32_t m, int32_t *len_host, double *norm_host, int64_t *eidx_host,
                              int64_t *node_host, double *ts_host, double *w_host)
{
    if (!h || m < 0 || m >= h->M) return ZT_ERR_ARG;
    ZT_HIP(hipDeviceSynchronize());
    const size_t N = (size_t)h->N, k = (size_t)h->k, rg = (size_t)h->rg;
    std::vector<u64> g(N * rg);
    ZT_HIP(hipMemcpy(g.data(), h->rows + (size_t)m * N * rg, N * rg * sizeof(u64), hipMemcpyDeviceToHost));
    for (size_t v = 0; v < N; ++v)
        decode_row(g.data() + v * rg, k, &len_host[v], &norm_host[v], eidx_host + v * k, node_host + v * k,
                   ts_host + v * k, w_host + v * k);
    return ZT_OK;
}

extern "C" int zt_tppr_export_rows(zt_tppr *h, int32_t m, const int64_t *ids_host, int64_t n, int32_t *len_host,
                                   double *norm_host, int64_t *eidx_host, int64_t *node_host, double *ts_host,
                                   double *w_host)
{
    if (!h || m < 0 || m >= h->M || n < 0 || (n > 0 && !ids_host)) return ZT_ERR_ARG;
    if (n == 0) return ZT_OK;
    for (int64_t q = 0; q < n; ++q)
        if (ids_host[q] < 0 || ids_host[q] >= h->N) { set_error("zt_tppr_export_rows: id out of range"); return ZT_ERR_RANGE; }
    ZT_HIP(hipDeviceSynchronize());
    const size_t k = (size_t)h->k, rg = (size_t)h->rg;
    long long *ids_dev = nullptr;
    u64 *buf_dev = nullptr;
    ZT_HIP(hipMalloc(&ids_dev, (size_t)n * sizeof(long long)));
    ZT_HIP(hipMalloc(&buf_dev, (size_t)n * rg * sizeof(u64)));
    ZT_HIP(hipMemcpy(ids_dev, ids_host, (size_t)n * sizeof(long long), hipMemcpyHostToDevice));
    k_gather_rows<<<(unsigned)n, 128>>>(h->rows + (size_t)m * (size_t)h->N * rg, ids_dev, n, (int)rg, buf_dev);
    std::vector<u64> g((size_t)n * rg);
    hipError_t e = hipMemcpy(g.data(), buf_dev, (size_t)n * rg * sizeof(u64), hipMemcpyDeviceToHost);
    (void)hipFree(ids_dev); (void)hipFree(buf_dev);
    ZT_HIP(e);
    for (size_t v = 0; v < (size_t)n; ++v)
        decode_row(g.data() + v * rg, k, &len_host[v], &norm_host[v], eidx_host + v * k, node_host + v * k,
                   ts_host + v * k, w_host + v * k);
    return ZT_OK;
}

namespace {
// the reference's dictionary items (iteration order) -> granules of one row, tag 0
int encode_row(const zt_tppr *h, u64 *r, int32_t len, double norm, const int64_t *eidx, const int64_t *node,
               const double *ts, const double *w)
{
    const size_t k = (size_t)h->k;
    if (len < 0 || len > (int)k) { set_error("zt_tppr_import: bad length"); return ZT_ERR_ARG; }
    u64 nb;
    memcpy(&nb, &norm, 8);
    r[0] = (u64)(unsigned)len; r[1] = (u64)(unsigned)nb; r[2] = nb >> 32;
    for (size_t j = 0; j < (size_t)len; ++j) {
        if (eidx[j] < 0 || eidx[j] > 0x7fffffffll || node[j] < 0 || node[j] >= h->N) {
            set_error("zt_tppr_import: id out of range");
            return ZT_ERR_RANGE;
        }
        u64 tb, wb;
        memcpy(&tb, &ts[j], 8);
        memcpy(&wb, &w[j], 8);
        u64 *e = r + HDR + j;
        e[0] = (u64)node[j]; e[k] = (u64)eidx[j];
        e[2 * k] = (u64)(unsigned)tb; e[3 * k] = tb >> 32;
        e[4 * k] = (u64)(unsigned)wb; e[5 * k] = wb >> 32;
    }
    return ZT_OK;
}

__global__ void k_scatter_tppr_rows(u64 *__restrict__ rows, const long long *__restrict__ ids, long long n, int rg,
                                    const u64 *__restrict__ in)
{
    const long long r = blockIdx.x;
    if (r >= n) return;
    u64 *dst = rows + ids[r] * rg;
    for (int c = threadIdx.x; c < rg; c += blockDim.x) dst[c] = in[r * rg + c];
}
}  // namespace

extern "C" int zt_tppr_import(zt_tppr *h, int32_t m, const int32_t *len_host, const double *norm_host,
                              const int64_t *eidx_host, const int64_t *node_host, const double *ts_host,
                              const double *w_host)
{
    if (!h || m < 0 || m >= h->M) return ZT_ERR_ARG;
    const size_t N = (size_t)h->N, k = (size_t)h->k, rg = (size_t)h->rg;
    std::vector<u64> g(N * rg, 0ull);
    for (size_t v = 0; v < N; ++v) {
        int rc = encode_row(h, g.data() + v * rg, len_host[v], norm_host[v], eidx_host + v * k, node_host + v * k,
                            ts_host + v * k, w_host + v * k);
        if (rc != ZT_OK) return rc;
    }
    ZT_HIP(hipDeviceSynchronize());
    ZT_HIP(hipMemcpy(h->rows + (size_t)m * N * rg, g.data(), N * rg * sizeof(u64), hipMemcpyHostToDevice));
    h->set[0].valid = h->set[1].valid = false;
    return ZT_OK;
}

extern "C" int zt_tppr_import_rows(zt_tppr *h, int32_t m, const int64_t *ids_host, int64_t n, const int32_t *len_host,
                                   const double *norm_host, const int64_t *eidx_host, const int64_t *node_host,
                                   const double *ts_host, const double *w_host)
{
    if (!h || m < 0 || m >= h->M || n < 0 || (n > 0 && !ids_host)) return ZT_ERR_ARG;
    if (n == 0) return ZT_OK;
    const size_t k = (size_t)h->k, rg = (size_t)h->rg;
    std::vector<u64> g((size_t)n * rg, 0ull);
    for (size_t v = 0; v < (size_t)n; ++v) {
        if (ids_host[v] < 0 || ids_host[v] >= h->N) { set_error("zt_tppr_import_rows: id out of range"); return ZT_ERR_RANGE; }
        int rc = encode_row(h, g.data() + v * rg, len_host[v], norm_host[v], eidx_host + v * k, node_host + v * k,
                            ts_host + v * k, w_host + v * k);
        if (rc != ZT_OK) return rc;
    }
    ZT_HIP(hipDeviceSynchronize());
    long long *ids_dev = nullptr;
    u64 *buf_dev = nullptr;
    ZT_HIP(hipMalloc(&ids_dev, (size_t)n * sizeof(long long)));
    ZT_HIP(hipMalloc(&buf_dev, (size_t)n * rg * sizeof(u64)));
    hipError_t e = hipMemcpy(ids_dev, ids_host, (size_t)n * sizeof(long long), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(buf_dev, g.data(), (size_t)n * rg * sizeof(u64), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_scatter_tppr_rows<<<(unsigned)n, 128>>>(h->rows + (size_t)m * (size_t)h->N * rg, ids_dev, n, (int)rg, buf_dev);
        e = hipDeviceSynchronize();
    }
    (void)hipFree(ids_dev); (void)hipFree(buf_dev);
    ZT_HIP(e);
    h->set[0].valid = h->set[1].valid = false;
    return ZT_OK;
}

