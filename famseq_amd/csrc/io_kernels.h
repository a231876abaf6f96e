// io_kernels.h — launchers of the streaming input/output stages (io_kernels.hip).
#ifndef FAMSEQ_IO_KERNELS_H_
#define FAMSEQ_IO_KERNELS_H_

#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace famseq {

constexpr uint16_t kPlMissing = 0xFFFF;  // all three PLs 0xFFFF = sample missing at this site
constexpr int kPlLutSize = 4096;         // pow(10,-k/10) is exactly 0 from k = 3240 on

hipError_t launch_unpack_pl16(const uint16_t *d_pl, const int32_t *d_col_of_member, const double *d_lut, int n_members,
                              int n_seq, int64_t n_sites, double *d_lk, hipStream_t stream);
// bare elementwise kernel of the posterior kernels' traffic shape (diagnostic; see io_kernels.hip)
hipError_t launch_stream_probe(const double *d_in, double *d_out1, double *d_out2, int64_t n_doubles, hipStream_t stream);
hipError_t launch_phred_call(const double *d_post, const double *d_single, const uint8_t *d_status,
                             const int32_t *d_seq_members, int n_members, int n_seq, int64_t n_sites, double *d_gpp,
                             double *d_fpp, int8_t *d_fgt, hipStream_t stream);

}  // namespace famseq
#endif
