// Shared between bank.hip and exchange.hip: the device-visible reservoir descriptor and the bank object.
#pragma once
#include <utility>
#include <vector>

#include "common.h"

namespace sml {

struct ResDesc {
    int n, d, n_model, n_out, n_aug, n_aug_pad, nslices, loaded;
    const int *slice_off;      // [nslices+1] first entry of each 64-row slice (entries are width*64 per slice)
    const unsigned short *sell_col;   // column into [x ; u] as uint16 (n + d <= 65535): 10 B per stored nonzero
    const double *sell_val;
    const int *perm;           // device position -> original row (kept for diagnostics; the kernels do not need it)
    const unsigned char *row_len;     // nonzeros of the row at each device position (<= 255)
    double *x[2];              // ping-pong state
    const double *wout;        // [n_out][n_aug_pad] row-major, zero padded
    // Compact copies, present when every value survives a round trip through float (weights read from the reference's NetCDF files are
    // NF90_REAL, src/mod_io.f90): the same numbers in half the bytes, converted back exactly on the fly.  NULL otherwise.
    const float *wout32;       // [n_out][n_aug_pad32], rows padded to whole 128-byte lines
    const float *sell_val32;   // as sell_val
    int n_aug_pad32, pad_;
    const double *mean, *stdv;
    const int *out_stat;       // [n_out] slot into mean/std, <0 = leave as is
    double leak;
};

struct HostRes {
    std::vector<void *> allocs;
    ResDesc desc{};
    uint64_t update_bytes = 0, readout_bytes = 0;
    uint64_t update_bytes32 = 0, readout_bytes32 = 0;      // the same accounting with 4-byte values (compact copies)
    float *wout32_alloc = nullptr;                         // the device buffer behind desc.wout32 (kept when the copy is withdrawn)
    std::vector<int> order;      // device position -> original row of the state vector
};

}  // namespace sml

struct sml_bank {
    int capacity = 0, max_d = 0, max_n_model = 0, max_n_out = 0;
    int ncu = 0;                    // compute units of the device this bank lives on (launch shapes follow it)
    int cur = 0;
    int max_nd = 0;                 // LDS doubles needed by k_update
    int max_n_out_loaded = 1;       // largest n_out among loaded slots (sizes the readout grid)
    std::vector<sml::HostRes> res;
    sml::ResDesc *d_descs = nullptr;
    double *d_feedback = nullptr, *d_local_model = nullptr, *d_outvec = nullptr, *d_partial = nullptr;
    double *d_vp = nullptr;            // [capacity][max_n_out] physics-model block of the readout (sml_bank_outvec_contribs), made on first use
    unsigned *d_counter = nullptr;     // work counter of the persistent readout
    bool descs_dirty = true;
    bool timing = false;
    bool allow_compact = true;      // sml_bank_use_compact(bank, 0): keep to the 8-byte copies whatever the weights are
    int compact = -1;               // 1: every loaded reservoir has compact copies and the predict kernels read those (decided when the descriptors are synchronised)
    bool timing_update = true;      // sml_bank_timing(b, 2): events around the readout only (the roofline kernel), none around the update
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_update, ev_readout;
    std::vector<std::pair<double *, size_t>> train_states;     // per slot: the training pass's states buffer, kept between passes
};

namespace sml {
int bank_sync_descs(sml_bank *b);
int comm_agree_min(sml_comm *c, int mine, int *agreed);
int exchange_egress(sml_exchange *ex, const double *fields_out_dev, const double *tisr_slice_dev, double *g_dev, double *f_dev, hipStream_t st);
// CU-masked streams go through a registry whose exit handler destroys the ones still alive (see bank.hip)
int masked_stream_create(hipStream_t *out, const uint32_t *mask, int nwords);
int masked_stream_destroy(hipStream_t st);
}
