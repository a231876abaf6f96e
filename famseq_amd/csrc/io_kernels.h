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

// GPP / FPP / FGT of n_items = n_sites * n_seq (site, sample) pairs -> the text the drivers print for each,
// "g0,g1,g2:f0,f1,f2:0/1\t", one kTextStride-byte record per pair: characters from byte 0, their number in the last byte
// (file.cpp:696-745: `ostream << double`, i.e. printf's %g; csrc/g6_core.h).  d_text must be 16-byte aligned.
constexpr int kTextStride = 80;
hipError_t launch_text_call(const double *d_gpp, const double *d_fpp, const int8_t *d_fgt, int64_t n_items, char *d_text,
                            hipStream_t stream);
// diagnostic / test aid: n doubles -> n 16-byte records (characters of g6_phred from byte 0, their number in byte 15)
hipError_t launch_g6_probe(const double *d_in, int64_t n, char *d_out, hipStream_t stream);

}  // namespace famseq
#endif
