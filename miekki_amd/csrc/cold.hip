// The column codec of the rows that do not fit the GPU.
//
// The reference keeps a collection larger than fast memory by deflating every column after the build
// (compress_index, Miekki.cpp:863-868, called from main.cpp:198; utils.cpp:321-360) and inflating a column whenever a
// batch needs it (get_minimizers, Miekki.cpp:881-898); its README expects most of the gain from the ORDER of the
// genomes ("A clever ordering of the lines could allow a very efficient column compression", README.md:136-138).  Here
// the rows beyond the matrix's HBM budget live in page-locked host memory and are streamed through HBM once per query
// chunk at the PCIe rate (api.hip); this file packs those rows so that fewer bytes cross PCIe.
//
// What there is to gain was measured first (tools/column_entropy.py, profiles/r4_column_entropy_strains.txt): genomes that
// are unrelated leave nothing (order-0 entropy 5.75 of 8 bits and no context helps: 1.39:1 at best, 1.35:1 for the
// reference's zlib), but STRAINS of one species next to each other in the list repeat most fingerprints of the genome
// before: at 0.1 % divergence 88 % of a row's fingerprints equal their left neighbour, at 1 % 42 %.  So the codec is the
// one a GPU decodes at memory speed: per piece of 1,024 genomes of a row, one bit per genome "differs from the genome
// before" (the first of a piece always does) and the differing fingerprints as they are -- 4.1:1 and 1.43:1 on those
// collections at one byte, 5.4:1 and 1.51:1 at two.  A row that would not shrink is stored as it is (independent genomes:
// every row), so packing never costs more than four bytes per row.
//
// Lifecycle, as in the reference: mk_index_compress after the build, everything that needs the rows as they are
// (appends, export, import, growing) unpacks first (need_raw_cold), queries stage a cold range by copying its PACKED bytes
// and expanding them in HBM (stage_cold_rows): the copy shrinks by the ratio, the scan is unchanged.
//
// Row format (16-byte aligned in the arena): u32 off[npieces + 1] (byte offsets of the pieces from the row's start;
// off[0] == 0xffffffff: the row follows as it is, G x W bytes), then per piece: 128 bytes of bits (bit j of 16-bit word l =
// genome 16 l + j of the piece), then the differing fingerprints (W bytes each, device byte order), padded to 4 bytes.
#include <cstring>

#include "mk_internal.hpp"

namespace mk {

namespace {

constexpr uint32_t kPiece = 1024;                  // genomes per piece
constexpr uint32_t kRawRow = 0xffffffffu;

template <int W> using fpw_t = typename std::conditional<W == 1, uint8_t, uint16_t>::type;

// the lane's 16 fingerprints of its piece (genomes past G repeat the last one: "same as before", never a literal)
template <int W>
__device__ __forceinline__ void load16(const uint8_t *__restrict__ row, uint32_t g0, uint32_t G, uint32_t (&fp)[16])
{
    const fpw_t<W> *__restrict__ r = reinterpret_cast<const fpw_t<W> *>(row);
#pragma unroll
    for (uint32_t j = 0; j < 16; ++j) fp[j] = g0 + j < G ? (uint32_t)r[g0 + j] : 0xffffffffu;
}

template <int W>
__device__ __forceinline__ uint32_t differ_bits(const uint32_t (&fp)[16], uint32_t g0, uint32_t G, uint32_t lane)
{
    uint32_t prev = (uint32_t)__shfl_up((int)fp[15], 1);           // the genome before this lane's first
    uint32_t bits = 0;
#pragma unroll
    for (uint32_t j = 0; j < 16; ++j) {
        const bool in = g0 + j < G;
        const bool d = in && ((lane == 0 && j == 0) || fp[j] != prev);
        bits |= (d ? 1u : 0u) << j;
        if (in) prev = fp[j];
    }
    return bits;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint32_t wave_excl(uint32_t v, uint32_t lane)
{
    uint32_t incl = v;
    for (uint32_t o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
    }
    return incl - v;
}

// bytes of every (row, piece): grid (npieces, nrows), one wave each
template <int W>
__global__ __launch_bounds__(64) void pack_count_kernel(const uint8_t *__restrict__ raw, uint64_t ld, uint32_t G, uint32_t npieces,
                                                        uint32_t *__restrict__ sizes)
{
    const uint32_t piece = blockIdx.x, row = blockIdx.y, lane = threadIdx.x;
    const uint32_t g0 = piece * kPiece + lane * 16u;
    uint32_t fp[16];
    load16<W>(raw + (uint64_t)row * ld, g0, G, fp);
    const uint32_t n = wave_sum((uint32_t)__popc(differ_bits<W>(fp, g0, G, lane)));
    if (lane == 0) sizes[(uint64_t)row * npieces + piece] = 128u + ((n * W + 3u) & ~3u);
}

// bytes of every row: header + pieces, or the row as it is when that is no larger; rounded up to 16
__global__ void pack_rowsize_kernel(const uint32_t *__restrict__ sizes, uint32_t nrows, uint32_t npieces, uint64_t raw_bytes,
                                    uint64_t *__restrict__ row_bytes)
{
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrows) return;
    uint64_t sum = 4ull * (npieces + 1);
    for (uint32_t p = 0; p < npieces; ++p) sum += sizes[(uint64_t)row * npieces + p];
    const uint64_t as_is = 4ull + ((raw_bytes + 3) & ~3ull);
    row_bytes[row] = ((sum < as_is ? sum : as_is) + 15) & ~15ull;
}

// write the rows: grid (npieces, nrows), one wave each; row_off[row] = the row's place in `out`
template <int W>
__global__ __launch_bounds__(64) void pack_write_kernel(const uint8_t *__restrict__ raw, uint64_t ld, uint32_t G, uint32_t npieces,
                                                        const uint32_t *__restrict__ sizes, const uint64_t *__restrict__ row_off,
                                                        uint8_t *__restrict__ out)
{
    const uint32_t piece = blockIdx.x, row = blockIdx.y, lane = threadIdx.x;
    const uint32_t g0 = piece * kPiece + lane * 16u;
    const uint8_t *__restrict__ src = raw + (uint64_t)row * ld;
    uint8_t *__restrict__ dst = out + row_off[row];
    // where this piece starts, and whether the row is stored as it is (the same rule as pack_rowsize_kernel)
    uint32_t before = 0, total = 0;
    for (uint32_t p = lane; p < npieces; p += 64) {
        const uint32_t s = sizes[(uint64_t)row * npieces + p];
        total += s;
        if (p < piece) before += s;
    }
    before = wave_sum(before); total = wave_sum(total);
    const uint64_t hdr = 4ull * (npieces + 1), raw_bytes = (uint64_t)G * W;
    const bool as_is = hdr + total >= 4ull + ((raw_bytes + 3) & ~3ull);
    uint32_t fp[16];
    load16<W>(src, g0, G, fp);
    if (as_is) {
        if (piece == 0 && lane == 0) *reinterpret_cast<uint32_t *>(dst) = kRawRow;
        fpw_t<W> *__restrict__ o = reinterpret_cast<fpw_t<W> *>(dst + 4);
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j)
            if (g0 + j < G) o[g0 + j] = (fpw_t<W>)fp[j];
        return;
    }
    const uint32_t poff = (uint32_t)hdr + before;
    if (lane == 0) {
        reinterpret_cast<uint32_t *>(dst)[piece] = poff;
        if (piece + 1 == npieces) reinterpret_cast<uint32_t *>(dst)[npieces] = (uint32_t)hdr + total;
    }
    const uint32_t bits = differ_bits<W>(fp, g0, G, lane);
    reinterpret_cast<uint16_t *>(dst + poff)[lane] = (uint16_t)bits;
    uint32_t at = wave_excl((uint32_t)__popc(bits), lane);
    fpw_t<W> *__restrict__ lit = reinterpret_cast<fpw_t<W> *>(dst + poff + 128);
#pragma unroll
    for (uint32_t j = 0; j < 16; ++j)
        if ((bits >> j) & 1u) lit[at++] = (fpw_t<W>)fp[j];
}

// expand packed rows into a staging buffer: grid (npieces, nrows), one wave each.  z = the packed bytes of rows
// [r0, r0 + nrows) as they lie in the arena from zoff[0] on; row r goes to dst + r * ld
template <int W>
__global__ __launch_bounds__(64) void unpack_kernel_cold(const uint8_t *__restrict__ z, const uint64_t *__restrict__ zoff, uint32_t G,
                                                         uint32_t npieces, uint8_t *__restrict__ dst, uint64_t ld)
{
    const uint32_t piece = blockIdx.x, row = blockIdx.y, lane = threadIdx.x;
    const uint8_t *__restrict__ src = z + (zoff[row] - zoff[0]);
    const uint32_t g0 = piece * kPiece + lane * 16u;
    fpw_t<W> *__restrict__ out = reinterpret_cast<fpw_t<W> *>(dst + (uint64_t)row * ld);
    const uint32_t word0 = *reinterpret_cast<const uint32_t *>(src);
    uint32_t v[16];
    if (word0 == kRawRow) {
        const fpw_t<W> *__restrict__ r = reinterpret_cast<const fpw_t<W> *>(src + 4);
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) v[j] = g0 + j < G ? (uint32_t)r[g0 + j] : 0u;
    } else {
        const uint32_t poff = reinterpret_cast<const uint32_t *>(src)[piece];
        const uint32_t bits = reinterpret_cast<const uint16_t *>(src + poff)[lane];
        const fpw_t<W> *__restrict__ lit = reinterpret_cast<const fpw_t<W> *>(src + poff + 128);
        uint32_t at = wave_excl((uint32_t)__popc(bits), lane);      // literals before this lane's
        uint32_t cur = (bits & 1u) || at == 0 ? 0u : (uint32_t)lit[at - 1];   // the genome before this lane's first (a piece's first bit is set)
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) {
            if ((bits >> j) & 1u) cur = (uint32_t)lit[at++];
            v[j] = cur;
        }
    }
    // (the padding of a row past its last genome takes whatever repeats; never past the pitch)
    const uint64_t byte0 = (uint64_t)g0 * W;
    if (byte0 + 16u * W <= ld) {
        if (W == 1) {
            uint4 o;
            o.x = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);     o.y = v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24);
            o.z = v[8] | (v[9] << 8) | (v[10] << 16) | (v[11] << 24);   o.w = v[12] | (v[13] << 8) | (v[14] << 16) | (v[15] << 24);
            *reinterpret_cast<uint4 *>(out + g0) = o;
        } else {
            uint4 o0, o1;
            o0.x = v[0] | (v[1] << 16); o0.y = v[2] | (v[3] << 16); o0.z = v[4] | (v[5] << 16); o0.w = v[6] | (v[7] << 16);
            o1.x = v[8] | (v[9] << 16); o1.y = v[10] | (v[11] << 16); o1.z = v[12] | (v[13] << 16); o1.w = v[14] | (v[15] << 16);
            reinterpret_cast<uint4 *>(out + g0)[0] = o0;
            reinterpret_cast<uint4 *>(out + g0)[1] = o1;
        }
    } else {
        for (uint32_t j = 0; j < 16; ++j)
            if (byte0 + (uint64_t)(j + 1) * W <= ld) out[g0 + j] = (fpw_t<W>)v[j];
    }
}

uint32_t pieces_of(const mk_ctx *c) { return std::max<uint32_t>(1, (c->G + kPiece - 1) / kPiece); }

// rows per round of the (un)packing passes: a few hundred MB of rows in HBM at a time
uint32_t chunk_rows(const mk_ctx *c) { return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(c->P - c->P_hot, (512ull << 20) / std::max<uint64_t>(c->ld, 1))); }

int launch_unpack(mk_ctx *c, const uint8_t *d_z, const uint64_t *d_zoff, uint32_t nrows, uint8_t *d_dst, hipStream_t st)
{
    if (!nrows) return MK_OK;
    const dim3 grid(pieces_of(c), nrows);
    if (c->W == 1) hipLaunchKernelGGL(unpack_kernel_cold<1>, grid, dim3(64), 0, st, d_z, d_zoff, c->G, pieces_of(c), d_dst, c->ld);
    else hipLaunchKernelGGL(unpack_kernel_cold<2>, grid, dim3(64), 0, st, d_z, d_zoff, c->G, pieces_of(c), d_dst, c->ld);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

}  // namespace

// ---- packing: two passes over the cold rows (sizes, then bytes) ------------------------------------------------------
int pack_cold(mk_ctx *c, uint64_t *raw_out, uint64_t *packed_out)
{
    if (raw_out) *raw_out = 0;
    if (packed_out) *packed_out = 0;
    if (!c->h_M || c->h_Z || !c->G) return MK_OK;                   // nothing cold, or packed already
    const uint32_t ncold = c->P - c->P_hot, npieces = pieces_of(c), per = chunk_rows(c);
    uint8_t *d_raw = nullptr, *d_out = nullptr;
    uint32_t *d_sizes = nullptr;
    uint64_t *d_rb = nullptr, *d_off = nullptr;
    auto drop = [&] { (void)hipFree(d_raw); (void)hipFree(d_out); (void)hipFree(d_sizes); (void)hipFree(d_rb); (void)hipFree(d_off); };
    const uint64_t raw_row = (uint64_t)c->G * c->W, worst_row = (4 + ((raw_row + 3) & ~3ull) + 15) & ~15ull;
    if (hipMalloc((void **)&d_raw, (uint64_t)per * c->ld) != hipSuccess || hipMalloc((void **)&d_out, (uint64_t)per * worst_row) != hipSuccess ||
        hipMalloc((void **)&d_sizes, (uint64_t)per * npieces * 4) != hipSuccess || hipMalloc((void **)&d_rb, (uint64_t)per * 8) != hipSuccess ||
        hipMalloc((void **)&d_off, (uint64_t)(per + 1) * 8) != hipSuccess) {
        (void)hipGetLastError();
        drop();
        set_error("out of device memory while packing the cold rows");
        return MK_ERR_NOMEM;
    }
    std::vector<uint64_t> zoff(ncold + 1, 0), rb(per), off(per + 1);
    int rc = MK_OK;
    auto count = [&](uint32_t r0, uint32_t n) -> bool {
        if (hipMemcpyAsync(d_raw, c->h_M + (uint64_t)r0 * c->ld, (uint64_t)n * c->ld, hipMemcpyHostToDevice, c->stream) != hipSuccess) return false;
        const dim3 grid(npieces, n);
        if (c->W == 1) hipLaunchKernelGGL(pack_count_kernel<1>, grid, dim3(64), 0, c->stream, d_raw, c->ld, c->G, npieces, d_sizes);
        else hipLaunchKernelGGL(pack_count_kernel<2>, grid, dim3(64), 0, c->stream, d_raw, c->ld, c->G, npieces, d_sizes);
        hipLaunchKernelGGL(pack_rowsize_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_sizes, n, npieces, raw_row, d_rb);
        return hipMemcpyAsync(rb.data(), d_rb, (uint64_t)n * 8, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
               hipStreamSynchronize(c->stream) == hipSuccess;
    };
    for (uint32_t r0 = 0; r0 < ncold && rc == MK_OK; r0 += per) {       // pass 1: how large
        const uint32_t n = std::min(per, ncold - r0);
        if (!count(r0, n)) { rc = MK_ERR_DEVICE; break; }
        for (uint32_t i = 0; i < n; ++i) zoff[r0 + i + 1] = zoff[r0 + i] + rb[i];
    }
    const uint64_t raw_total = (uint64_t)ncold * c->ld, packed_total = zoff[ncold];
    if (raw_out) *raw_out = raw_total;
    if (packed_out) *packed_out = packed_total;
    // not worth it (independent genomes: every row is stored as it is): the rows stay as they are
    if (rc == MK_OK && (double)packed_total > 0.95 * (double)((uint64_t)ncold * raw_row)) { drop(); if (packed_out) *packed_out = raw_total; return MK_OK; }
    uint8_t *hz = nullptr;
    if (rc == MK_OK && hipHostMalloc((void **)&hz, std::max<uint64_t>(packed_total, 16), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        set_error("the packed cold rows (%llu bytes) do not fit page-locked host memory beside the raw ones", (unsigned long long)packed_total);
        rc = MK_ERR_NOMEM;
    }
    for (uint32_t r0 = 0; r0 < ncold && rc == MK_OK; r0 += per) {       // pass 2: the bytes
        const uint32_t n = std::min(per, ncold - r0);
        if (!count(r0, n)) { rc = MK_ERR_DEVICE; break; }
        for (uint32_t i = 0; i <= n; ++i) off[i] = zoff[r0 + i] - zoff[r0];
        bool ok = hipMemcpyAsync(d_off, off.data(), (uint64_t)(n + 1) * 8, hipMemcpyHostToDevice, c->stream) == hipSuccess;
        const dim3 grid(npieces, n);
        if (ok) {
            if (c->W == 1) hipLaunchKernelGGL(pack_write_kernel<1>, grid, dim3(64), 0, c->stream, d_raw, c->ld, c->G, npieces, d_sizes, d_off, d_out);
            else hipLaunchKernelGGL(pack_write_kernel<2>, grid, dim3(64), 0, c->stream, d_raw, c->ld, c->G, npieces, d_sizes, d_off, d_out);
            ok = hipMemcpyAsync(hz + zoff[r0], d_out, off[n], hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
        }
        if (!ok) rc = MK_ERR_DEVICE;
    }
    drop();
    if (rc == MK_OK) {
        uint64_t *dz = nullptr;
        if (hipMalloc((void **)&dz, (uint64_t)(ncold + 1) * 8) != hipSuccess ||
            hipMemcpy(dz, zoff.data(), (uint64_t)(ncold + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(dz); rc = MK_ERR_DEVICE; }
        else {
            (void)hipHostFree(c->h_M);
            c->h_M = nullptr;
            c->h_Z = hz; c->z_bytes = packed_total; c->d_zoff = dz; c->h_zoff.swap(zoff);
            hz = nullptr;
        }
    }
    if (hz) (void)hipHostFree(hz);
    if (rc == MK_ERR_DEVICE) set_error("packing the cold rows failed: %s", hipGetErrorString(hipGetLastError()));
    return rc;
}

// ---- back to the rows as they are (appends, export, import, growth) ---------------------------------------------------
int need_raw_cold(mk_ctx *c)
{
    if (!c->h_Z) return MK_OK;
    MK_HIP(hipStreamSynchronize(c->stream));
    MK_HIP(hipStreamSynchronize(c->copy_stream));
    const uint32_t ncold = c->P - c->P_hot, per = chunk_rows(c);
    uint8_t *hm = nullptr, *d_z = nullptr, *d_raw = nullptr;
    if (hipHostMalloc((void **)&hm, (uint64_t)ncold * c->ld, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        set_error("the cold rows (%llu bytes unpacked) do not fit page-locked host memory", (unsigned long long)ncold * c->ld);
        return MK_ERR_NOMEM;
    }
    uint64_t zmax = 0;
    for (uint32_t r0 = 0; r0 < ncold; r0 += per) zmax = std::max(zmax, c->h_zoff[std::min(ncold, r0 + per)] - c->h_zoff[r0]);
    int rc = MK_OK;
    if (hipMalloc((void **)&d_z, std::max<uint64_t>(zmax, 16)) != hipSuccess || hipMalloc((void **)&d_raw, (uint64_t)per * c->ld) != hipSuccess) rc = MK_ERR_NOMEM;
    for (uint32_t r0 = 0; r0 < ncold && rc == MK_OK; r0 += per) {
        const uint32_t n = std::min(per, ncold - r0);
        bool ok = hipMemcpyAsync(d_z, c->h_Z + c->h_zoff[r0], c->h_zoff[r0 + n] - c->h_zoff[r0], hipMemcpyHostToDevice, c->stream) == hipSuccess;
        ok = ok && hipMemsetAsync(d_raw, 0, (uint64_t)n * c->ld, c->stream) == hipSuccess;
        if (ok) rc = launch_unpack(c, d_z, c->d_zoff + r0, n, d_raw, c->stream);
        ok = ok && rc == MK_OK && hipMemcpyAsync(hm + (uint64_t)r0 * c->ld, d_raw, (uint64_t)n * c->ld, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
             hipStreamSynchronize(c->stream) == hipSuccess;
        if (!ok && rc == MK_OK) rc = MK_ERR_DEVICE;
    }
    (void)hipFree(d_z); (void)hipFree(d_raw);
    if (rc != MK_OK) { (void)hipHostFree(hm); if (rc != MK_ERR_NOMEM) set_error("unpacking the cold rows failed: %s", hipGetErrorString(hipGetLastError())); else set_error("out of device memory while unpacking the cold rows"); return rc; }
    (void)hipHostFree(c->h_Z); (void)hipFree(c->d_zoff);
    for (int b = 0; b < 2; ++b) { if (c->d_zstage[b]) (void)hipFree(c->d_zstage[b]); c->d_zstage[b] = nullptr; }
    c->zstage_cap = 0;
    c->h_Z = nullptr; c->d_zoff = nullptr; c->z_bytes = 0; c->h_zoff.clear();
    c->h_M = hm;
    return MK_OK;
}

// ---- queries: cold rows [r_lo, r_hi) (matrix row numbers) into `d_dst` (row r_lo first), on stream st, through packed
// staging buffer `b` -- the raw copy of api.hip when the rows are not packed
int stage_cold_rows(mk_ctx *c, uint64_t r_lo, uint64_t r_hi, uint8_t *d_dst, int b, hipStream_t st)
{
    if (r_hi <= r_lo) return MK_OK;
    if (!c->h_Z) {
        MK_HIP(hipMemcpyAsync(d_dst, c->h_M + (r_lo - c->P_hot) * c->ld, (r_hi - r_lo) * c->ld, hipMemcpyHostToDevice, st));
        return MK_OK;
    }
    const uint64_t i0 = r_lo - c->P_hot, i1 = r_hi - c->P_hot, bytes = c->h_zoff[i1] - c->h_zoff[i0];
    if (bytes > c->zstage_cap || !c->d_zstage[b]) {
        // (sized once for the largest window any caller stages: the worst packed size of a stage's worth of rows)
        set_error("packed staging buffer too small");
        return MK_ERR_STATE;
    }
    MK_HIP(hipMemcpyAsync(c->d_zstage[b], c->h_Z + c->h_zoff[i0], bytes, hipMemcpyHostToDevice, st));
    return launch_unpack(c, c->d_zstage[b], c->d_zoff + i0, (uint32_t)(i1 - i0), d_dst, st);
}

// the packed staging buffers, sized for `rows` rows at their worst (called wherever d_cold_stage is (re)sized)
int ensure_zstage(mk_ctx *c, uint64_t rows)
{
    if (!c->h_Z) return MK_OK;
    const uint64_t raw_row = (uint64_t)c->G * c->W, worst_row = (4 + ((raw_row + 3) & ~3ull) + 15) & ~15ull, need = rows * worst_row;
    if (c->d_zstage[0] && c->d_zstage[1] && need <= c->zstage_cap) return MK_OK;
    MK_HIP(hipStreamSynchronize(c->stream));
    MK_HIP(hipStreamSynchronize(c->copy_stream));
    for (int b = 0; b < 2; ++b) {
        if (c->d_zstage[b]) (void)hipFree(c->d_zstage[b]);
        c->d_zstage[b] = nullptr;
        MK_HIP(hipMalloc((void **)&c->d_zstage[b], std::max<uint64_t>(need, 16)));
    }
    c->zstage_cap = need;
    return MK_OK;
}

}  // namespace mk
