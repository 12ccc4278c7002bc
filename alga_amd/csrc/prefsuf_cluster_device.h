// alga_amd/csrc/prefsuf_cluster_device.h -- device-side definitions of the clustered minimizer join shared by prefsuf_cluster.hip (one
// GPU: the index and the source-side probe kernels) and prefsuf_shard.hip (N GPUs: the bucket-sharded join): k-mer order, cluster key,
// sort key of a target, directory record -> entry slice.
#pragma once
#include <hip/hip_runtime.h>
#include "prefsuf_common.h"
#include "../../include/alga_amd.h"
#include "prefsuf_device.h"

namespace alga {

// order hash of a k-mer (lo: nucleotides 0..15, hi: 16..31, both masked to the k-mer's length)
__device__ __forceinline__ uint32_t kmer_hash(uint32_t lo, uint32_t hi) {
    // ONE multiply (32-bit integer multiplies run at quarter rate).  A second mixing round made no measurable difference to the
    // minimizers' statistics (runs per node, entries per source, sources the pair kernel finishes): what has to be well mixed is
    // the CLUSTER key, and that gets its own mix, once per run.
    return (lo ^ __funnelshift_l(hi, hi, 13) ^ (hi >> 7)) * 0x9E3779B1u;
}

// ORDER of the k-mers of a window: the smallest order key is the window's minimizer.
//   bit 31      content class: 0 for a k-mer that starts with A followed by C or G (one k-mer in eight), 1 for any other
//   bits 30..8  23 bits of the order hash
//   bits 7..0   position in the read (ties to the left; equal windows of a source and a target agree on it)
// The class bit is what makes k_node_runs cheap: a window of w = 64 k-mers holds a class-0 k-mer with probability 1 - (7/8)^64,
// so the minimizer of (nearly) every window is a class-0 k-mer, and those are found with a few word-parallel bit operations on
// the 2-bit rows -- the order hash is evaluated for one k-mer position in eight instead of all of them (13.6 G hashes and
// 11.4 ms per build at the north-star size in round 2).  It stays a function of the window's content alone, which is all the
// join needs: a source window and the equal target prefix choose the same k-mer.  Windows without a class-0 k-mer take the
// minimum over their class-1 k-mers (brute force: the prefix window inside k_node_runs, a source's other windows in
// k_probe_clustered's slow path).
// nucleotide codes: A = 0, C = 1, G = 2, T = 3 (include/Params.h:275-279); nucleotide j of a k-mer = bits (2j, 2j + 1) of lo
__device__ __forceinline__ bool kmer_class0(uint32_t lo) { return (lo & 3u) == 0u && ((((lo >> 2) ^ (lo >> 3)) & 1u) != 0u); }
__device__ __forceinline__ uint32_t order_key0(uint32_t h, int pos) { return ((h >> 1) & 0x7FFFFF00u) | (uint32_t) pos; }      // of a class-0 k-mer
__device__ __forceinline__ uint32_t order_key(uint32_t h, uint32_t lo, int pos) { return order_key0(h, pos) | (kmer_class0(lo) ? 0u : 0x80000000u); }
// class-0 positions among the 16 nucleotides of `cur` (bit 2j: nucleotide j); nxt = the following row word
__device__ __forceinline__ uint32_t class0_mask16(uint32_t cur, uint32_t nxt) {
    const uint32_t nx = __funnelshift_r(cur, nxt, 2);      // nucleotide j + 1 at bits (2j, 2j + 1)
    return ~(cur | (cur >> 1)) & (nx ^ (nx >> 1)) & 0x55555555u;
}
// bits 0, 2, 4, ... 30 of x (the others are zero) -> bits 0 .. 15
__device__ __forceinline__ uint32_t compress_even(uint32_t x) {
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    return (x | (x >> 8)) & 0x0000FFFFu;
}

// Cluster key of a minimizer = a second, bijective mix of its order hash.  The order hashes of MINIMIZERS are minima of w
// uniform values -- concentrated near zero -- so bucketing the entry array by their own top bits would put most clusters in
// 1/w of the buckets; the mix spreads them evenly.  The BUCKET bits (above the m_C field, tgt_sort_key below) are never all ones:
// the sort key of a target then never is 0xFFFFFFFF ("not a target") whatever its m_C, m_C can be read back from it, and a sort
// on the bucket bits alone already puts every target before every non-target (the last bucket stays empty).
__device__ __forceinline__ uint32_t cluster_key(uint32_t h, int fs) {
    uint32_t k = h * 0x9E3779B1u;                          // (one multiply instead of three: +2 % entries scanned, +28 % deferred sources)
    k ^= k >> 15;
    k *= 0x85EBCA77u;
    k ^= k >> 13;
    k *= 0xC2B2AE3Du;
    return (k >> (fs + CL_MBITS)) == (0xFFFFFFFFu >> (fs + CL_MBITS)) ? k ^ 0x80000000u : k;
}

// Sort key of a target: the cluster key with the CL_MBITS bits right below its bucket bits replaced by m_C, the position of the
// minimizer in the target's prefix (field at bit fs = idx_shift - CL_MBITS).  Inside a bucket the entries are therefore ordered
// by m_C >> 3 first (the sort stops there: launch_cluster_store), and the directory (k_tgt_dir) knows where every eighth of that order starts: a source run that covers the
// windows [p0, p1) with its minimizer at q can only match targets with q - p1 < m_C <= q - p0 and reads that slice of the bucket
// alone -- the other entries of the cluster are the reads of the same locus that start too far left or right of the run's
// windows (half of them at 30x coverage).  0xFFFFFFFF stays reserved for "not a target".
__device__ __forceinline__ uint32_t tgt_sort_key(uint32_t ckey, uint32_t m_c, int fs) {
    const uint32_t fm = ((1u << CL_MBITS) - 1u) << fs;
    return (ckey & ~fm) | ((m_c << fs) & fm);               // never all ones: cluster_key
}
__device__ __forceinline__ bool same_cluster(uint32_t entry_key, uint32_t ckey, int fs) {
    return ((entry_key ^ ckey) & ~(((1u << CL_MBITS) - 1u) << fs)) == 0u;
}

// directory record of a bucket: {first entry, entries, first entry (relative, saturating bytes) with m_C >> 3 >= 0..3, >= 4..7}
// -> the entries [e0, e0 + cnt) a run {q | p0 << 8 | p1 << 16} has to look at
// k_pile_build / k_pile_probe / k_probe_stream: more than one bucket in this many irregular and the pairwise kernels take the build (prefsuf_pile.hip)
constexpr unsigned long long PILE_IRREGULAR_ONE_IN = ALGA_PILE_IRREGULAR_ONE_IN;      // include/alga_amd.h
__device__ __forceinline__ void run_slice(const uint4 &rec, uint32_t run_y, uint32_t &e0, uint32_t &cnt) {
    const int q = (int) (run_y & 255u), p0 = (int) ((run_y >> 8) & 255u), p1 = (int) ((run_y >> 16) & 255u);
    int mlo = q - p1 + 1, mhi = q - p0;
    mlo = mlo < 0 ? 0 : mlo; mhi = mhi > 63 ? 63 : mhi;
    e0 = rec.x; cnt = rec.y;
    if (mhi < mlo) { cnt = 0u; return; }
    if (rec.y > 255u) return;                              // offsets saturate: the whole bucket
    const int s0 = mlo >> 3, s1 = (mhi >> 3) + 1;
    const uint32_t b0 = ((s0 < 4 ? rec.z : rec.w) >> (8 * (s0 & 3))) & 255u;
    const uint32_t b1 = s1 >= 8 ? rec.y : (((s1 < 4 ? rec.z : rec.w) >> (8 * (s1 & 3))) & 255u);
    e0 = rec.x + b0; cnt = b1 - b0;
}

// k-mer starting at nucleotide i of a 2-bit row (words readable up to index (2i >> 5) + 2): its hash and its packed
// order key (class | 23-bit order | position); the smallest key of a window is the window's minimizer
__device__ __forceinline__ void kmer_key(const uint32_t *row, int i, bool valid, const ClusterCfg &cc, uint32_t &h, uint32_t &pk) {
    const int bit = 2 * i, q = bit >> 5, r = bit & 31;
    const uint32_t x0 = row[q], x1 = row[q + 1], x2 = row[q + 2];
    const uint32_t lo = funnel(x0, x1, r) & cc.lo_mask;
    h = kmer_hash(lo, funnel(x1, x2, r) & cc.hi_mask);
    pk = valid ? order_key(h, lo, i) : 0xFFFFFFFFu;
}

__device__ __forceinline__ uint32_t bperm(uint32_t v, int src_lane) { return (uint32_t) __builtin_amdgcn_ds_bpermute(src_lane << 2, (int) v); }

} // namespace alga
