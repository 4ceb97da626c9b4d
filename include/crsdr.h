/*
 * crsdr.h -- C ABI of the MI355X-native coherent-alignment engine (libcrsdr.so).
 *
 * Drop-in boundary for ONE path of mlaaks/coherent-rtlsdr: the ccoherent/cdsp DSP path
 * (int8 IQ -> complex float -> zero-padded FFT cross-correlation against the reference-noise
 * row -> argmax lag + magnitude -> phase estimate -> rotate (+shift) -> re-quantise into the
 * N x L receive matrix that cpacketize publishes).  Everything else of the reference (USB,
 * tuner control, console, ZMQ sockets) stays where it is.
 *
 * Two levels, mirroring the reference's own plug point (class cdsp is "wraps to volk kernels.
 * In future, these could be mapped to custom code", src/cdsp.cc:19; the RASPBERRYPI build
 * already swaps the FFT backend behind fft_scheme, include/cdsp.h:23-32):
 *   (i)  per-op entry points, one per cdsp static method   -> unit parity, drop-in for cdsp.cc
 *   (ii) a batched plan replacing ccoherent::{ctor,queuelag,computelag} plus the per-row
 *        csdrdevice::{convtofloat,est_phasecorrect,phasecorrect} and cpacketize::write chain
 *        (src/ccoherent.cc:245-294), because per-op launches would be launch-bound.
 *
 * Conventions (reference units): blocksize B = int8 values per row per block = FFT length in
 * complex points; L = B/2 complex samples per row; row 0 = reference-noise channel.
 * Complex arrays are interleaved float (re,im) -- layout-identical to std::complex<float>,
 * fftwf_complex and lv_32fc_t (src/ccoherent.cc:65).
 * Every function returns 0 on success or a negative CRSDR_E* code; nothing throws across the
 * ABI; crsdr_last_error() gives the text for the calling thread's last failure.
 * There is NO CPU fallback: without a usable HIP device every compute entry point fails with
 * CRSDR_ENODEV.
 */
#ifndef CRSDR_H
#define CRSDR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRSDR_ABI_VERSION 1

enum {
    CRSDR_OK = 0,
    CRSDR_EINVAL = -1,  /* bad argument (NULL, non power-of-two size, out of range) */
    CRSDR_ENOMEM = -2,  /* host or device allocation failed */
    CRSDR_EHIP = -3,    /* a HIP runtime call or kernel launch failed */
    CRSDR_ENODEV = -4,  /* no HIP device available */
    CRSDR_ESTATE = -5   /* call sequence error (fetch before submit, ...) */
};

int crsdr_abi_version(void);
const char *crsdr_last_error(void);
int crsdr_device_count(int *count);
/* Page-locked host memory for the rows / packet buffers handed to crsdr_plan_submit / crsdr_plan_fetch with
 * CRSDR_MEM_HOST -- the role volk_malloc / fftwf_alloc_complex play in the reference (src/ccoherent.cc:44-47,
 * 66-69: aligned allocations owned by the engine).  Pageable memory works too, at roughly half the PCIe rate
 * and with host-blocking copies. */
int crsdr_host_alloc(void **ptr, size_t bytes);
int crsdr_host_free(void *ptr);

/* Name, compute units, engine / memory clocks (kHz) and memory size of a device, for benchmark records
 * ("GPU clocks recorded").  Any output pointer may be NULL; name is NUL-terminated in name_cap bytes. */
int crsdr_device_info(int device, char *name, int name_cap, int *compute_units, int *clock_khz, int *memory_clock_khz,
                      size_t *memory_bytes);

/* ------------------------------------------------------------------------------------------
 * (i) per-op entry points == class cdsp (include/cdsp.h:36-71).  Host pointers in and out;
 *     each call stages through device memory on the current device (default stream) and
 *     returns when the result is in `out`.  Note the reference's mixed units for n.
 * ---------------------------------------------------------------------------------------- */

/* cdsp::convtosigned(const uint8_t*,const uint8_t*,int)  include/cdsp.h:40, src/cdsp.cc:21-34
 * out[i] = in[i] ^ 0x80, n bytes (n % 8 == 0 like the reference's 64-bit loop). */
int crsdr_convtosigned(const uint8_t *in, uint8_t *out, int n);

/* cdsp::convtofloat(const float*,const int8_t*,int) and the complex overload
 * include/cdsp.h:41,45, src/cdsp.cc:36-44.  n = number of int8 values (= floats written). */
int crsdr_convtofloat(float *out, const int8_t *s8bit, int n);

/* cdsp::scalarmul  include/cdsp.h:47, src/cdsp.cc:46-49.  n = complex count. */
int crsdr_scalarmul(float *out, const float *in, float scalar_re, float scalar_im, int n);

/* cdsp::convto8bit  include/cdsp.h:43, src/cdsp.cc:51-54.  n = complex count (2n bytes out). */
int crsdr_convto8bit(int8_t *out, const float *in, int n);

/* cdsp::conj_dotproduct  include/cdsp.h:49, src/cdsp.cc:61-66.  res[0..1] = sum a*conj(b). */
int crsdr_conj_dotproduct(float *res, const float *a, const float *b, int n);

/* cdsp::magsquared  include/cdsp.h:61, src/cdsp.cc:100-103.  n = complex count. */
int crsdr_magsquared(float *out, const float *in, int n);

/* cdsp::conjugatemul  include/cdsp.h:63, src/cdsp.cc:105-108.  out = in1 * conj(in2). */
int crsdr_conjugatemul(float *out, const float *in1, const float *in2, int n);

/* cdsp::indexofmax(float*,int)  include/cdsp.h:68, src/cdsp.cc:135-139: first strict maximum. */
int crsdr_indexofmax(uint32_t *index, const float *in, int n);

/* cdsp::fft(out,in,fft_scheme*)  include/cdsp.h:65, src/cdsp.cc:110-120, with the plan geometry
 * of src/ccoherent.cc:78-93 passed explicitly instead of an fftwf_plan: `howmany` contiguous
 * transforms of n complex points (dist n, stride 1), sign -1 = FFTW_FORWARD, +1 =
 * FFTW_BACKWARD, unnormalised, out of place.  n: power of two, 16 <= n <= 16384 (one
 * LDS-resident transform per workgroup; longer transforms exist only inside the plan, where the
 * four-step order never has to be undone). */
int crsdr_fft(float *out, const float *in, int n, int sign, int howmany);

/* ------------------------------------------------------------------------------------------
 * (ii) batched plan == ccoherent + per-row csdrdevice DSP members + cpacketize::write
 * ---------------------------------------------------------------------------------------- */

enum {
    CRSDR_MODE_FAITHFUL = 0, /* reference order of operations: lag reported, samples not
                                shifted (the lag feeds the resampler servo, src/ccontrol.cc:93-119),
                                phase estimated at zero offset */
    CRSDR_MODE_DIGITAL = 1   /* north_star "rotate/shift": row k is shifted by its lag, the
                                phase is estimated over the overlap, then rotate + quantise */
};

enum {
    CRSDR_MEM_HOST = 0,   /* pointer is host memory: the plan copies H2D / D2H */
    CRSDR_MEM_DEVICE = 1  /* pointer is device memory on the plan's device */
};

/* crsdr_plan_submit flags */
enum {
    CRSDR_REFNOISE_ENABLED = 1u << 0, /* crefnoise::isenabled(), gate of src/ccoherent.cc:271 */
    CRSDR_OFFSET_BINARY = 1u << 1,    /* input is librtlsdr's raw uint8: fuse cdsp::convtosigned
                                         (x ^ 0x80, include/common.h:114-122) into the loads */
    CRSDR_INPUT_READY = 1u << 2,      /* device input is already complete: the reference-spectrum
                                         kernel need not wait for earlier work on the stream */
    CRSDR_NO_LAG = 1u << 3            /* no row requests a lag this block ("locked" steady state,
                                         src/ccontrol.cc:117-120): phase path only, no FFT */
};

typedef struct crsdr_plan crsdr_plan;

typedef struct crsdr_plan_desc {
    int32_t nrows;      /* N: reference row + signal rows in the receive matrix (hdr0::N) */
    int32_t blocksize;  /* B: int8 values per row per block; power of two, 16..4194304.  B <= 16384:
                           one LDS-resident transform per row; B > 16384 (config 5: 2^21): four-step
                           B = (B/16384) x 16384 transform streamed through HBM, max_batch = 1 */
    int32_t mode;       /* CRSDR_MODE_* */
    int32_t device;     /* HIP device ordinal */
    int32_t row_begin;  /* first signal row this plan owns (>= 1); 0 = default (1) */
    int32_t row_count;  /* signal rows owned; 0 = default (all: nrows - row_begin).  Multi-GPU:
                           rank g owns a contiguous slab, the ref row is replicated */
    int32_t max_batch;  /* most consecutive blocks one crsdr_plan_submit_batch may carry (1..64);
                           0 = default (1).  Sizes every per-block device buffer. */
    uint32_t reserved;
} crsdr_plan_desc;

/* ccoherent::ccoherent (src/ccoherent.cc:32-95): allocates every device buffer, twiddle table,
 * stream and event; nothing is allocated on the per-block path afterwards. */
int crsdr_plan_create(crsdr_plan **plan, const crsdr_plan_desc *desc);
/* ccoherent::~ccoherent (src/ccoherent.cc:97-112) */
int crsdr_plan_destroy(crsdr_plan *plan);
/* back to construction state: phasecorr = phasecorrprev = 1+0j, lag = 0 (src/csdrdevice.cc:36-40) */
int crsdr_plan_reset(crsdr_plan *plan);

/* Run the plan's kernels on a caller-owned hipStream_t (e.g. the stream an RCCL gather is
 * enqueued on); NULL restores the plan's own stream. */
int crsdr_plan_set_stream(crsdr_plan *plan, void *hip_stream);

/* One ccoherent::threadf iteration (src/ccoherent.cc:245-294), asynchronous.
 *   rows      [nrows][B] int8 (uint8 with CRSDR_OFFSET_BINARY), row 0 = reference; a plan that
 *             owns a slab reads only row 0 and its own rows.
 *   mem_kind  CRSDR_MEM_HOST (copied) or CRSDR_MEM_DEVICE (read in place; must stay valid
 *             until the next fetch/sync).
 *   readcnt   [nrows] host array copied into the packet (cpacketize::write, src/cpacketizer.cc:
 *             142,163), or NULL: every row gets `seq`.
 *   lag_mask  [nrows] host bytes, non-zero = csdrdevice::is_lagrequested() (src/ccoherent.cc:266);
 *             NULL = every owned signal row.  Unlike the reference there is no nfft = 8 cap
 *             (src/ccoherent.cc:124): every requested row is cross-correlated in this block.
 *   seq       hdr0::globalseqn for this block (src/cpacketizer.cc:113). */
int crsdr_plan_submit(crsdr_plan *plan, const void *rows, int mem_kind, const uint32_t *readcnt,
                      const uint8_t *lag_mask, uint32_t seq, uint32_t flags);

/* nblocks consecutive ccoherent::threadf iterations in one submit: the same results as nblocks
 * crsdr_plan_submit calls in a row (the EMA phasor and the last lag are carried from block to
 * block), but every kernel is launched once over nblocks x rows workgroups -- host cost and
 * launch gaps are paid per batch, and a GPU that owns a small slab still fills its 256 CUs.
 *   rows          block t at rows + t * block_stride (bytes; 0 = nrows * B, i.e. contiguous)
 *   readcnt       [nblocks][nrows] or NULL (block t gets seq + t)
 *   lag_mask      [nrows], applies to every block of the batch
 *   seq           hdr0::globalseqn of block 0; block t gets seq + t
 * Packets land at device_packet + t * packet_stride. */
int crsdr_plan_submit_batch(crsdr_plan *plan, const void *rows, int mem_kind, int nblocks, size_t block_stride,
                            const uint32_t *readcnt, const uint8_t *lag_mask, uint32_t seq, uint32_t flags);

/* Wait for the last submitted block and copy results to host arrays (any may be NULL):
 *   lag    [nrows] int32  idx - L, what csdrdevice::set_lag receives (src/ccoherent.cc:232)
 *   mag    [nrows] float  sqrt(peak / L)                         (src/ccoherent.cc:204)
 *   frac   [nrows] float  3-point parabolic peak offset in samples (extra output; the reference
 *                         computes and discards its own variant, src/ccoherent.cc:206-219)
 *   phasor [nrows][2]     csdrdevice::get_phasecorrect() -- the port-5557 debug payload
 *                         (src/cpacketizer.cc:127,131-134); entry 0 is 0
 *   packet crsdr_plan_packet_bytes() bytes: hdr0{seq,N,L,0} + u32 readcnt[N] + int8 [N][B]
 *          (include/cpacketizer.h:32-37; parser matlabclient/zmqsdr.c:118-144).
 * Rows outside the plan's slab hold whatever the bound buffer held (zeros by default). */
int crsdr_plan_fetch(crsdr_plan *plan, int32_t *lag, float *mag, float *frac, float *phasor,
                     int8_t *packet);
/* Same for block `block` (0-based) of the last submitted batch; -1 = its last block, which is
 * what crsdr_plan_fetch returns. */
int crsdr_plan_fetch_block(crsdr_plan *plan, int block, int32_t *lag, float *mag, float *frac, float *phasor,
                           int8_t *packet);

/* Fractional-delay correction (BASELINE config 5 "long-block regime + fractional-delay phase correction"; SURVEY 8 note
 * "Fractional delay": report D and, if applied, apply it as a linear phase ramp in the frequency domain).  The reference
 * computes a 3-point estimate and discards it (src/ccoherent.cc:206-219); its authors' study of applying one
 * (matlabclient/notes.m:9-40) finds a signal-dependent gain between estimate and true delay, hence `gain` / the override.
 * CRSDR_MODE_DIGITAL only; off by default (the reference-faithful behaviour).  Every block size: long blocks (blocksize > 16384) run a
 * second four-step pass (below), LDS-resident blocks one extra kernel per batch behind the phase kernels (row -> forward transform ->
 * x the response -> inverse -> int8 over the row the phase kernel wrote), for which enable = 1 and 2 are the same and nothing is allocated.
 * With it on, row k of the matrix is the row advanced by lag_k + D_k samples -- a circular advance of the zero-padded row in
 * the frequency domain: X[f] * exp(+2 pi i f_s (lag_k + D_k) / B), f_s the signed bin index -- then rotated by the phasor
 * and quantised like cdsp::convto8bit; for D = 0 that is the digital mode's zero-filled integer shift.
 *   D_k = frac_override[k] if given (host [nrows], copied; entry 0 ignored), else gain * frac_k (this block's estimate).
 *   Bounds: |gain| <= 128 and |frac_override[k]| <= 64 samples (CRSDR_EINVAL otherwise): the response's phase is formed in fp32 and
 *   keeps 1e-5 rad up to there; a proper peak's estimate is |frac| <= 1/2.  Larger delays belong in the integer lag.
 * lag / mag / frac / phasor outputs are unchanged (the phase is still estimated on the integer-aligned row).
 *   enable = 1: allocates (once, here -- never on the per-block path) a second cf32 work area of 8 * blocksize bytes per owned row,
 *               so that the correction pass reuses the correlation pass's first stage, and 128 KiB per owned row for the rows'
 *               response spectra; if the work area cannot be had the pass repeats its first stage instead (= enable 2);
 *   enable = 2: the memory-lean form: no second work area, the pass repeats its first stage (same results, bit for bit);
 *   enable = 0: off; both buffers are freed (after a sync). */
int crsdr_plan_set_frac_apply(crsdr_plan *plan, int enable, float gain, const float *frac_override);

/* Pipelined fetch of the LAST submitted batch into page-locked host memory (crsdr_host_alloc): the device-to-host copies
 * run on the plan's copy stream as soon as the batch's kernels have finished -- while the NEXT submit's host-to-device copies
 * use the other direction of the link; the next submit's kernels wait on the device for these copies, so the plan's output
 * buffers are never overwritten under them.  The host engine's loop (src/ccoherent.cc:245-294 run a batch at a time):
 *     submit_batch(b) ; fetch_batch_async(b) ; submit_batch(b+1) ; fetch_batch_async(b+1) ; fetch_wait() -> batch b has landed ; ...
 * Any pointer may be NULL.  lag / mag / frac: [nblocks][nrows]; phasor [nblocks][nrows][2]; packets: the batch's nblocks
 * packets, packet t at packets + t * host_packet_stride (>= crsdr_plan_packet_bytes).  Pageable destinations work but
 * serialise.  crsdr_plan_fetch_wait blocks until the OLDEST outstanding asynchronous fetch has landed (at most four may be
 * outstanding; later ones keep flying) and reports a kernel-side error like crsdr_plan_fetch does. */
int crsdr_plan_fetch_batch_async(crsdr_plan *plan, int32_t *lag, float *mag, float *frac, float *phasor, int8_t *packets,
                                 size_t host_packet_stride);
int crsdr_plan_fetch_wait(crsdr_plan *plan);

/* Block until everything submitted so far has finished (no copies). */
int crsdr_plan_sync(crsdr_plan *plan);

size_t crsdr_plan_packet_bytes(const crsdr_plan *plan);   /* 16 + 4N + N*B */
size_t crsdr_plan_matrix_offset(const crsdr_plan *plan);  /* 16 + 4N */
size_t crsdr_plan_packet_stride(const crsdr_plan *plan);  /* bytes between the packets of a batch */

/* Device-resident results for pipelines that never leave HBM (multi-GPU gather, benchmarks).
 * The packet pointer is 4-byte aligned and the matrix inside it 256-byte aligned; per-row arrays
 * are [max_batch][nrows] (block t of the last batch at + t * nrows).  lag / mag / frac alternate between two
 * such arrays from submit to submit: call this after the submit whose results are wanted; the pointers stay valid
 * (and unmodified) until the submit after the next one. */
int crsdr_plan_device_buffers(crsdr_plan *plan, void **packet, void **lag, void **mag,
                              void **frac, void **phasor);
/* Write the packets of later submits into a caller-owned device buffer instead: block t of a
 * batch at device_packet + t * packet_stride (matrix start and stride 4-byte aligned, stride >=
 * packet_bytes); NULL restores the plan's own buffer.  Does not synchronise: double-buffering
 * against a gather in flight is the caller's business. */
int crsdr_plan_bind_packet(crsdr_plan *plan, void *device_packet, size_t packet_stride);

/* Sharded plans (row_begin / row_count set): slab output for the matrix exchange between GPUs.
 * With a slab bound, the owned rows of block t of a batch are written densely at
 *   device_slab + t*slab_stride + (row - row_begin)*blocksize          (slab_stride >= row_count*blocksize)
 * instead of into the packet matrix, so one rank's batch is a contiguous [nblocks][row_count][blocksize]
 * buffer -- the send buffer of ONE all-to-all per batch (equal splits: blocks [q*nblocks/G, (q+1)*nblocks/G) go
 * to rank q).  Header + readcnt + row 0 (cpacketize::write(0, ..), src/ccoherent.cc:253) are written only for
 * blocks hdr_first <= t < hdr_first + hdr_count -- the blocks this rank assembles -- into packet
 * (t - hdr_first) of the bound packet buffer.  device_slab = NULL returns to packet output.  No sync. */
int crsdr_plan_bind_slab(crsdr_plan *plan, void *device_slab, size_t slab_stride, int hdr_first, int hdr_count);

/* On the assembling rank: recv [nsrc][nblocks][(nrows-1)/nsrc][blocksize] int8 (the all-to-all's receive
 * buffer: chunk src = that rank's slabs of the nblocks blocks assembled here) -> matrix rows 1 + src*per ..
 * of packet j at device_packets + j*packet_stride.  Replaces the per-row cpacketize::write(c, ..) calls of
 * src/ccoherent.cc:277 for rows computed on other GPUs.  Device pointers; asynchronous on hip_stream
 * (a hipStream_t, NULL = the null stream).  Independent of any plan. */
int crsdr_assemble_slabs(void *device_packets, size_t packet_stride, int nrows, int blocksize, const void *device_recv, int nsrc,
                         int nblocks, void *hip_stream);

/* ---- exchange slots: rows + per-row scalars, one message per peer -------------------------------------
 * The gather of SURVEY 8e carries the int8 rows PLUS 24 bytes per row of {lag, mag, frac, phasor, readcnt}: what
 * csdrdevice::set_lag (src/ccoherent.cc:232-233), the port-5557 debug payload (src/cpacketizer.cc:127,131-134) and the packet
 * header's per-device read counters (src/cpacketizer.cc:142,163) consume on the assembling side -- a rank need only know the
 * counters of row 0 and of the rows it owns.  crsdr_plan_bind_slab_ex is crsdr_plan_bind_slab with a TAIL per block: block t of a
 * batch goes to the slot  device_slab + t*slab_stride =
 *     [row_count][blocksize] int8 rows | at +tail_offset:  int32 lag[rc] | float mag[rc] | float frac[rc] | float phasor[rc][2] | uint32 readcnt[rc]
 * (tail_offset >= row_count*blocksize, 4-byte aligned, tail_offset + 24*row_count <= slab_stride; 0 = no tail).  A rank's
 * slots of the blocks rooted on one peer are consecutive, so they travel as ONE message whatever the block count. */
int crsdr_plan_bind_slab_ex(crsdr_plan *plan, void *device_slab, size_t slab_stride, int hdr_first, int hdr_count, size_t tail_offset);

/* Slot geometry both sides of an exchange agree on (pure arithmetic, no device needed):
 *   per = (nrows-1)/nranks rows per rank;  tail_offset = per*blocksize;  slot_stride = tail_offset + 24*per rounded up to 16;
 *   scalars_stride = 20*nrows rounded up to 16 (one assembled scalars block per packet, layout below). */
int crsdr_exchange_geometry(int nrows, int blocksize, int nranks, size_t *slot_stride, size_t *tail_offset, size_t *scalars_stride);

/* Which rank assembles block t of a batch of nblocks blocks: blocks are dealt in runs of bpr = ceil(nblocks / nranks),
 * root(t) = t / bpr -- rank q assembles blocks [q*bpr, min(nblocks, (q+1)*bpr)), a rotating root at batch granularity (a fixed
 * root would ingest (G-1)/G of every block over its own xGMI links).  first/count: rank's range (count may be 0). */
int crsdr_exchange_rooted_blocks(int nblocks, int nranks, int rank, int *first, int *count);

/* On the assembling rank: recv [nsrc][nblocks][slot_stride] (chunk src = rank src's slots of the nblocks blocks assembled here,
 * what an all-to-all of the send buffers delivers) ->  rows into matrix rows 1 + src*per .. of packet j at
 * device_packets + j*packet_stride;  tails into the scalars block j at device_scalars + j*scalars_stride =
 *     int32 lag[nrows] | float mag[nrows] | float frac[nrows] | float phasor[nrows][2]      (row 0: zeros, like crsdr_plan_fetch;
 *     the phasor part is the N x complex<float> port-5557 payload as is)
 * and, whenever the slots have tails, the read counters of rank src's rows into packet j's header (uint32 readcnt[N] at +16): the header
 * the assembling rank's own plan wrote holds ITS view of every counter, right for row 0 and its own rows.
 * self_rank >= 0 with device_self != NULL: that rank's chunk is read from device_self (its own send slots) instead of recv.
 * device_scalars may be NULL.  Asynchronous on hip_stream.  Replaces crsdr_assemble_slabs when tails travel with the rows. */
int crsdr_assemble_slots(void *device_packets, size_t packet_stride, void *device_scalars, size_t scalars_stride, int nrows, int blocksize,
                         const void *device_recv, int nsrc, int nblocks, size_t slot_stride, size_t tail_offset, int self_rank,
                         const void *device_self, void *hip_stream);

/* ---- the exchange itself under the C ABI: RCCL over xGMI, one process per GPU --------------------------------------
 * librccl is resolved at run time (dlopen "librccl.so.1"): a host that never creates an exchange does not need it.
 * The reference has no collective (its only transport is ZMQ, src/cpacketizer.cc:58-66); this is the one exchange step
 * SURVEY 8e adds.  Rendez-vous: rank 0 calls crsdr_exchange_unique_id and hands the 128 bytes to the other ranks by
 * whatever channel the host has (MPI, a socket, a file); every rank then calls crsdr_exchange_create. */
#define CRSDR_EXCHANGE_ID_BYTES 128
typedef struct crsdr_exchange crsdr_exchange;
int crsdr_exchange_unique_id(void *id /* [CRSDR_EXCHANGE_ID_BYTES] */);
int crsdr_exchange_create(crsdr_exchange **x, const void *id, int nranks, int rank, int device);
int crsdr_exchange_destroy(crsdr_exchange *x);

enum {
    CRSDR_XCHG_STAGED = 0,   /* one message per peer into device_recv, then crsdr_assemble_slots (what an all-to-all does) */
    CRSDR_XCHG_INPLACE = 1   /* rows land straight in the packet matrix (one message per block and peer, no assembly copy
                                of remote rows); only the 24 B/row tails go through a small staging buffer */
};
/* One batch of nblocks blocks (every rank calls it with the same nblocks / mode, after the submit that filled device_send on
 * hip_stream): rank q = root of blocks [q*bpr, ...) receives every rank's slots of those blocks; on return (stream order) the
 * count packets at device_packets hold all rows and device_scalars the assembled scalars blocks.
 *   device_send   this rank's slots [nblocks][slot_stride] (crsdr_plan_bind_slab_ex with the geometry above)
 *   device_recv   staging of >= nranks * bpr * slot_stride bytes (CRSDR_XCHG_STAGED; may be NULL for CRSDR_XCHG_INPLACE)
 * All RCCL calls are issued in one group on hip_stream; nothing blocks the host. */
int crsdr_exchange_batch(crsdr_exchange *x, int mode, const void *device_send, void *device_recv, int nblocks, void *device_packets,
                         size_t packet_stride, void *device_scalars, size_t scalars_stride, int nrows, int blocksize, void *hip_stream);

/* The point-to-point operations crsdr_exchange_batch issues for (nranks, rank, nblocks, mode), in issue order -- pure host
 * arithmetic, exported so that the matching of every send with its receive can be checked without GPUs (tests simulate the
 * ranks and pair each rank's sends to a peer with that peer's receives from it, first in first out, as RCCL does).
 * buffer: 0 = device_send, 1 = device_recv, 2 = device_packets, 3 = the exchange's tail staging ([nranks][bpr][24*per rounded to 16]). */
typedef struct crsdr_xop {
    int32_t peer;      /* the other rank */
    int32_t is_recv;   /* 0 = send, 1 = receive */
    int32_t buffer;    /* which buffer `offset` is relative to (above) */
    int32_t block;     /* batch-relative block index the bytes belong to (first block for multi-block messages) */
    uint64_t offset, bytes;
} crsdr_xop;
int crsdr_exchange_schedule(int nranks, int rank, int nblocks, int mode, int nrows, int blocksize, size_t packet_stride,
                            crsdr_xop *ops, int capacity, int *count);

/* ---- a sharded plan and its exchange as ONE engine: what a C++ host with one process per GPU runs (host/ccoherent.cc with
 * ranks > 1; the loop of src/ccoherent.cc:245-294 a batch at a time, rows split over the GPUs of the node).
 * crsdr_exchange_bind_plan allocates, once, everything the exchange of the plan's batches needs on the device -- two ring-buffered
 * sets of: the send slots [max_batch][slot_stride] the plan writes its rows and tails into, the receive staging, the
 * bpr = ceil(max_batch / nranks) packets this rank assembles per batch and their scalars blocks -- plus a side stream for the
 * exchange, so the host handles no device memory and no stream.  The plan must own rank's slab (row_begin = 1 + rank * per,
 * row_count = per) on the exchange's device.  Afterwards the plan is driven through the two calls below only. */
int crsdr_exchange_bind_plan(crsdr_exchange *x, crsdr_plan *plan, int mode /* CRSDR_XCHG_* */);
/* crsdr_plan_submit_batch (same arguments) into the next set, then crsdr_exchange_batch on the side stream: everything is enqueued
 * on return; the exchange of this batch runs under the compute of the next one.  At most two batches may be outstanding. */
int crsdr_exchange_submit_batch(crsdr_exchange *x, const void *rows, int mem_kind, int nblocks, size_t block_stride, const uint32_t *readcnt,
                                const uint8_t *lag_mask, uint32_t seq, uint32_t flags);
/* Waits for the OLDEST outstanding batch and copies what this rank assembled of it to host memory: blocks [*first, *first + *count)
 * of the batch (crsdr_exchange_rooted_blocks; count may be 0) -- packet j at packets + j * host_packet_stride, its scalars block
 * (int32 lag[N] | float mag[N] | float frac[N] | float phasor[N][2], 20 * nrows bytes) at scalars + j * host_scalars_stride.
 * own_tails: {lag, mag, frac, phasor} of this rank's OWN rows for every block of the batch -- block t at own_tails + t * host_tails_stride
 * as  int32 lag[per] | float mag[per] | float frac[per] | float phasor[per][2] | uint32 readcnt[per]  (24 * per bytes): what csdrdevice::set_lag needs on the
 * process that reads those dongles (src/ccoherent.cc:232-233), whichever rank assembles the block; *nblocks = blocks in the batch.
 * Any pointer may be NULL.  Reports a kernel-side error like crsdr_plan_fetch_wait does (the plan is rolled back and what else was
 * outstanding on THIS rank is dropped).  The ranks' exchange sequences stay matched -- the failing rank's sends of those batches were
 * issued -- but its peers assembled garbage rows from it: a host that wants the batches back resubmits them on every rank, in the
 * same order. */
int crsdr_exchange_fetch_rooted(crsdr_exchange *x, int8_t *packets, size_t host_packet_stride, void *scalars, size_t host_scalars_stride,
                                void *own_tails, size_t host_tails_stride, int *first, int *count, int *nblocks);

/* With profiling enabled (CRSDR_PROFILE_SUBMIT): elapsed GPU milliseconds on the plan's stream
 * between the start and the end of the most recent submit. */
int crsdr_plan_last_elapsed_ms(crsdr_plan *plan, float *ms);

/* Per-kernel timing with hipEvents recorded on the stream each kernel is launched on.
 * enable: keep event pairs for the last `slots` submits (0 disables) for the kernels in
 * `kernel_mask` (bit CRSDR_KERNEL_*; CRSDR_PROFILE_SUBMIT adds whole-submit start/stop events --
 * every recorded pair costs a few microseconds of stream time, so profile only what is read).
 * kernel_times: copy the durations (ms) of kernel `which` for the submits recorded since enable,
 * oldest first. */
enum { CRSDR_KERNEL_REF_SPECTRUM = 0, CRSDR_KERNEL_XCORR_LAG = 1, CRSDR_KERNEL_PHASE_DOT = 2, CRSDR_KERNEL_ALIGN_QUANT = 3 };
#define CRSDR_PROFILE_ALL_KERNELS 0xFu
#define CRSDR_PROFILE_SUBMIT (1u << 31)
int crsdr_plan_enable_profiling(crsdr_plan *plan, int slots, uint32_t kernel_mask);
int crsdr_plan_kernel_times(crsdr_plan *plan, int which, float *ms, int capacity, int *count);

/* ------------------------------------------------------------------------------------------
 * (iii) downstream helper: what the reference's beamformer computes first from a packet
 * ---------------------------------------------------------------------------------------- */

/* Sample covariance of the signal channels of one aligned receive matrix, as
 * beamformclient/heatmap2d2.cpp:189-199 forms it:  X(n, c) = (I + jQ)/127 of channel c >= 1 (row 0,
 * the reference channel, is dropped), per-channel mean removed,  Rxx = (1/L) X^H X.
 *   matrix [nrows][blocksize] int8 -- the data part of a packet (packet + crsdr_plan_matrix_offset)
 *   rxx    [(nrows-1)][(nrows-1)][2] float, row-major, rxx[a][b] = (1/L) sum_n conj(x_a[n]) x_b[n] - conj(mean_a) mean_b
 * Runs as an int8 GEMM on the matrix cores (v_mfma_i32_32x32x32_i8), exact integer sums (int32 per at most 65536 bytes of a row,
 * added in 64 bits: full-scale rows of any length are exact), fp64 epilogue.
 * blocksize % 32 == 0; above 65536: % 512 == 0.  mem_kind: CRSDR_MEM_HOST (copied) or CRSDR_MEM_DEVICE (both pointers on the device;
 * the matrix may sit where a packet holds it, 4-byte aligned). */
int crsdr_covariance(float *rxx, const int8_t *matrix, int nrows, int blocksize, int mem_kind);

/* Signal / noise subspaces of a Hermitian covariance, replacing noisesubspace(Rxx, K)
 * (beamformclient/heatmap2d2.cpp:69-79: BDCSVD of Rxx, Un = U.rightCols(M - K)).
 *   rxx [m][m][2] float row-major (the output of crsdr_covariance), 2 <= m <= 64
 *   sv  [m] singular values, descending (NULL to skip)
 *   vec [m][m][2] float row-major; column r is the singular vector of sv[r], so the noise subspace for K
 *       sources is columns K .. m-1 (a basis of it: only the projector Un Un^H is unique)
 * One workgroup, one-sided Jacobi in fp64 with the matrix resident in LDS; no host arithmetic.
 * Returns CRSDR_ESTATE if the iteration did not converge (vec / sv are still written). */
int crsdr_noisesubspace(float *vec, float *sv, const float *rxx, int m, int mem_kind);

/* 2-D MUSIC pseudo-spectrum of a uniform rectangular array, replacing pmusic2dvec(Un, d, Mx, My, Cx, Cy)
 * with s_vecd2d and pmusic (beamformclient/heatmap2d2.cpp:103-147, called at :199 with d = 1.225*1.24/3,
 * Mx = 7, My = 3, Cx = Cy = 100):
 *   a(alpha, beta)[iy*mx + ix] = exp(2 pi j ix d cos(alpha) sin(beta)) exp(2 pi j iy d cos(beta)),
 *   alpha = cx pi / ncx, beta = cy pi / ncy,   pm[cx][cy] = ( |a|^2 / |Un^H a|^2 )^2      (squared twice, :123)
 *   vec [m][m][2] as written by crsdr_noisesubspace, m = mx*my <= 64; Un = its columns k .. m-1, 1 <= k < m
 *   pm  [ncx][ncy] float row-major, not normalised (the reference divides by the maximum for plotting, :202-203) */
int crsdr_pmusic2d(float *pm, const float *vec, int m, int k, float d, int mx, int my, int ncx, int ncy, int mem_kind);

#ifdef __cplusplus
}
#endif
#endif /* CRSDR_H */
