/* mi355x_stream_compaction.h -- C ABI of the scan / stream-compaction library (part of libmi355x_pathtracer.so).
 *
 * Drop-in for the reference's stream_compaction/ library (file:line relative to the reference root):
 *
 *   StreamCompaction::CPU::scan                 stream_compaction/cpu.cu:20      sc_cpu_scan
 *   StreamCompaction::CPU::compactWithoutScan   stream_compaction/cpu.cu:39      sc_cpu_compact_without_scan
 *   StreamCompaction::CPU::compactWithScan      stream_compaction/cpu.cu:58      sc_cpu_compact_with_scan
 *   StreamCompaction::Naive::scan               stream_compaction/naive.cu:34    sc_naive_scan
 *   StreamCompaction::Efficient::scan           stream_compaction/efficient.cu:36 sc_efficient_scan
 *   StreamCompaction::Efficient::compact        stream_compaction/efficient.cu:79 sc_efficient_compact
 *   StreamCompaction::Thrust::scan              stream_compaction/thrust.cu:20   sc_thrust_scan
 *   StreamCompaction::Common::kernMapToBoolean  stream_compaction/common.cu:25   sc_map_to_boolean_device
 *   StreamCompaction::Common::kernScatter       stream_compaction/common.cu:40   sc_scatter_device
 *   <ns>::timer().getGpu/CpuElapsedTimeForPreviousOperation  common.h:48-132     sc_last_gpu_ms / sc_last_cpu_ms
 *
 * Same argument meaning as the reference: n elements, host pointers in and out (the GPU variants allocate and
 * copy internally, exactly like the reference's), exclusive prefix sum, compaction keeps non-zero elements in
 * order and returns their count.  The three GPU scan entry points are one implementation here (a single-pass chained
 * scan with decoupled look-back written for wave64; the reference's O(log n)-launch Naive and Blelloch variants are
 * teaching steps, not something to preserve) and give identical results.  The *_device forms take device
 * pointers and a hipStream_t and do no allocation or copy: that is what a production caller wants.
 * GPU entry points return 0 on success, non-zero (PTX_ERR_*) on failure with the message in ptx_last_error().
 */
#ifndef MI355X_STREAM_COMPACTION_H
#define MI355X_STREAM_COMPACTION_H

#ifdef __cplusplus
extern "C" {
#endif

void sc_cpu_scan(int n, int *odata, const int *idata);
int sc_cpu_compact_without_scan(int n, int *odata, const int *idata);
int sc_cpu_compact_with_scan(int n, int *odata, const int *idata);

int sc_naive_scan(int n, int *odata, const int *idata);
int sc_efficient_scan(int n, int *odata, const int *idata);
int sc_thrust_scan(int n, int *odata, const int *idata);
/* returns the number of elements kept, or -1 on failure */
int sc_efficient_compact(int n, int *odata, const int *idata);

/* device-pointer forms; workspace: sc_scan_workspace_bytes(n) bytes of device memory, 8-byte aligned, contents arbitrary;
 * the scan may run in place (d_odata == d_idata), the compaction may not */
unsigned long long sc_scan_workspace_bytes(int n);
int sc_scan_device(int n, int *d_odata, const int *d_idata, void *d_workspace, void *stream);
int sc_compact_device(int n, int *d_odata, const int *d_idata, int *d_count, void *d_workspace, void *stream);

/* the two building-block kernels of the reference's own compaction (common.h:38-41), on device arrays of n ints:
 * bools[i] = idata[i] != 0 ? 1 : 0;   and   bools[i] == 1  =>  odata[indices[i]] = idata[i].   Enqueued on `stream`, no sync. */
int sc_map_to_boolean_device(int n, int *d_bools, const int *d_idata, void *stream);
int sc_scatter_device(int n, int *d_odata, const int *d_idata, const int *d_bools, const int *d_indices, void *stream);

float sc_last_gpu_ms(void);      /* device time of the kernels of the previous GPU call (hipEvent), ms */
float sc_last_cpu_ms(void);      /* wall time of the previous sc_cpu_* call, ms */

int sc_ilog2(int x);             /* common.h:21-27 */
int sc_ilog2ceil(int x);         /* common.h:29-31 */

#ifdef __cplusplus
}
#endif
#endif
