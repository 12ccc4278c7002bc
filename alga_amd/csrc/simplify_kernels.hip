// alga_amd/csrc/simplify_kernels.hip -- first step of the graph simplifier on the GPU.
//
// Reference: the PrefSuf path of GraphSimplifier::simplifyGraphOld (src/GraphSimplifiers/GraphSimplifier.cpp:90-125) starts with
//   Graph::sortEdgesByIncreasingOffset          (src/DataStructures/Graph.cpp:584-614: lists ordered by (offset, neighbour))
//   GraphSimplifier::cutNonAndWeaklyMetricTriangles (src/GraphSimplifiers/GraphSimplifier.cpp:228-348): an edge i -> b of weight
//       w <= MAX_OFFSET_PARALLEL_PATHS is removed when the SHORTEST two-edge path i -> a -> b has exactly weight w; all decisions
//       are taken on the unchanged graph (first pass collects, second pass removes), so they are independent per edge.
// The edge list arrives from the overlap engine grouped by source and sorted by (neighbour, offset) -- a CSR whose rows can be
// binary-searched for the closing edge a -> b.  One thread per node: most nodes have ONE out-edge and can lose nothing (a
// two-edge path to their own neighbour would need a self-loop there), they copy their edge and leave; the rest order their
// list, decide every edge, and replay Graph::removeDirectedEdge's swap-with-last removals (src/DataStructures/Graph.cpp:96-119) so
// that the lists come out in the very order the reference leaves them in.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "prefsuf_kernels.h"
#include "simplify_kernels.h"

namespace alga {

// rowptr[t] = first edge with src >= t, t in [0, n]
__global__ void __launch_bounds__(256) k_edge_rowptr(const alga_edge_dev *__restrict__ e, uint64_t m, int32_t n, uint32_t *__restrict__ rowptr) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i <= m; i += (uint64_t) gridDim.x * blockDim.x) {
        const int64_t lo = i == 0 ? 0 : (int64_t) e[i - 1].src + 1;
        const int64_t hi = i == m ? (int64_t) n : (int64_t) e[i].src;
        for (int64_t t = lo; t <= hi; t++) rowptr[t] = (uint32_t) i;
    }
}

// weight of a -> b in the (neighbour, offset)-sorted row of a: the smallest one (rows hold one edge per neighbour after
// retainOnlySmallestOffset; with several the shortest two-edge path takes the smallest anyway); -1 = no such edge
__device__ __forceinline__ int32_t closing_weight(const alga_edge_dev *__restrict__ e, const uint32_t *__restrict__ rowptr, int32_t a, int32_t b) {
    uint32_t lo = rowptr[a], hi = rowptr[a + 1];
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (e[mid].dst < b) lo = mid + 1; else hi = mid; }
    return (lo < rowptr[a + 1] && e[lo].dst == b) ? e[lo].offset : -1;
}

__global__ void __launch_bounds__(256) k_cut_triangles(const alga_edge_dev *__restrict__ e, const uint32_t *__restrict__ rowptr, int32_t n, int32_t mopp,
                                                        alga_edge_dev *__restrict__ sorted, alga_edge_dev *__restrict__ lst, uint32_t *__restrict__ out_cnt,
                                                        unsigned long long *__restrict__ removed_total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b0 = rowptr[i], deg = rowptr[i + 1] - b0;
    alga_edge_dev *S = sorted + b0;                        // the row as sortEdgesByIncreasingOffset leaves it: walked in this order
    alga_edge_dev *L = lst + b0;                           // the same row while the removals permute it
    if (deg <= 1) { if (deg) L[0] = e[b0]; out_cnt[i] = deg; return; }
    for (uint32_t k = 0; k < deg; k++) {                   // (offset, neighbour); insertion sort, rows are short
        const alga_edge_dev x = e[b0 + k];
        uint32_t j = k;
        while (j > 0 && (S[j - 1].offset > x.offset || (S[j - 1].offset == x.offset && S[j - 1].dst > x.dst))) { S[j] = S[j - 1]; j--; }
        S[j] = x;
    }
    for (uint32_t k = 0; k < deg; k++) L[k] = S[k];
    uint32_t size = deg, removed = 0;
    for (uint32_t k = 0; k < deg; k++) {                   // decisions on the unchanged graph, in list order (GraphSimplifier.cpp:297-318)
        const alga_edge_dev x = S[k];
        if (x.offset > mopp) continue;                                       // long edges stay (:301-303)
        bool have = false;
        int32_t best = 0;
        for (uint32_t k2 = 0; k2 < deg; k2++) {                               // dst[b] = min over a of w(i, a) + w(a, b) (:283-295)
            const alga_edge_dev y = e[b0 + k2];
            const int32_t w2 = closing_weight(e, rowptr, y.dst, x.dst);
            if (w2 >= 0) { const int32_t d = y.offset + w2; if (!have || d < best) { best = d; have = true; } }
        }
        if (have && best == x.offset) {                                       // equal distances only (:310)
            // Graph::removeDirectedEdge(i, b): every entry with that neighbour, scanning from the back, swapped with the last
            int64_t p = (int64_t) size - 1;
            for (int64_t q = (int64_t) size - 1; q >= 0; q--)
                if (L[q].dst == x.dst) { const alga_edge_dev t = L[q]; L[q] = L[p]; L[p] = t; size--; p--; removed++; }
        }
    }
    out_cnt[i] = size;
    if (removed) atomicAdd(removed_total, (unsigned long long) removed);
}

__global__ void __launch_bounds__(256) k_compact_rows(const alga_edge_dev *__restrict__ work, const uint32_t *__restrict__ rowptr,
                                                       const uint32_t *__restrict__ out_cnt, const uint32_t *__restrict__ out_rowptr, int32_t n,
                                                       alga_edge_dev *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = out_cnt[i], s = rowptr[i], d = out_rowptr[i];
    for (uint32_t k = 0; k < c; k++) out[d + k] = work[s + k];
}

// ---- contig trimming (src/main.cpp:633-725) --------------------------------------------------
// rows[M + i] = reverse complement of rows[i] (MyUtils::getComplimentaryString(getReverse(.)), src/main.cpp:641-643): one thread per
// output word; nucleotide j of the result is 3 - nucleotide (len - 1 - j) of the contig
__global__ void __launch_bounds__(256) k_revcomp_rows(uint32_t *__restrict__ rows, int stride, int32_t *__restrict__ len, int32_t M) {
    const uint64_t t = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = t / (uint64_t) stride;
    const int wq = (int) (t - i * (uint64_t) stride);
    if (i >= (uint64_t) M) return;
    const int n = len[i];
    const uint32_t *src = rows + i * (size_t) stride;
    uint32_t v = 0;
    for (int j = 0; j < 16; j++) {
        const int pos = 16 * wq + j;
        if (pos < n) {
            const int q = n - 1 - pos;
            v |= (3u - ((src[q >> 4] >> ((q & 15) << 1)) & 3u)) << (2 * j);
        }
    }
    rows[((size_t) M + i) * (size_t) stride + wq] = v;
    if (wq == 0) len[(size_t) M + i] = n;
}

// trimLeft[d] = longest overlap |i| - offset over the edges i -> d between two FORWARD contigs (src/main.cpp:683-697)
__global__ void __launch_bounds__(256) k_trim_left(const alga_edge_dev *__restrict__ e, uint64_t m, const int32_t *__restrict__ len, int32_t M,
                                                    int32_t *__restrict__ trim) {
    for (uint64_t k = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; k < m; k += (uint64_t) gridDim.x * blockDim.x) {
        const alga_edge_dev x = e[k];
        if (x.src < M && x.dst < M) atomicMax(&trim[x.dst], len[x.src] - x.offset);
    }
}

void launch_revcomp_rows(uint32_t *rows, int stride, int32_t *len, int32_t M, hipStream_t s) {
    if (M <= 0) return;
    const uint64_t t = (uint64_t) M * (uint64_t) stride;
    hipLaunchKernelGGL(k_revcomp_rows, dim3((unsigned) ((t + 255) / 256)), dim3(256), 0, s, rows, stride, len, M);
}
void launch_trim_left(const alga_edge_dev *e, uint64_t m, const int32_t *len, int32_t M, int32_t *trim, hipStream_t s) {
    if (m == 0) return;
    hipLaunchKernelGGL(k_trim_left, dim3((unsigned) std::min<uint64_t>((m + 255) / 256, 4096)), dim3(256), 0, s, e, m, len, M, trim);
}

void launch_edge_rowptr(const alga_edge_dev *e, uint64_t m, int32_t n, uint32_t *rowptr, hipStream_t s) {
    hipLaunchKernelGGL(k_edge_rowptr, dim3((unsigned) std::min<uint64_t>((m + 256) / 256, 8192)), dim3(256), 0, s, e, m, n, rowptr);
}
void launch_cut_triangles(const alga_edge_dev *e, const uint32_t *rowptr, int32_t n, int32_t mopp, alga_edge_dev *sorted, alga_edge_dev *lst, uint32_t *out_cnt,
                          unsigned long long *removed_total, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_cut_triangles, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, e, rowptr, n, mopp, sorted, lst, out_cnt, removed_total);
}
void launch_compact_rows(const alga_edge_dev *work, const uint32_t *rowptr, const uint32_t *out_cnt, const uint32_t *out_rowptr, int32_t n,
                         alga_edge_dev *out, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_compact_rows, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, work, rowptr, out_cnt, out_rowptr, n, out);
}

} // namespace alga
