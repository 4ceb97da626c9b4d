// ccontrol.h -- the per-dongle lag servo of the reference (include/ccontrol.h:28-49, src/ccontrol.cc:25-123)
// as a closed-loop MODEL over the synthetic devices: the consumer of the hot path's lag
// output.  The reference slews the RTL2832 resampler by p = 2^-11 tanh(lag/100) for
// t = 0.9 |lag / (p fs)| seconds (rtlsdr_set_sample_freq_correction_f) and re-measures until
// |lag| <= sync_threshold, then marks the device synchronized -- which stops further lag requests,
// i.e. the engine falls into its "locked" cadence (src/ccontrol.cc:117-120, src/ccoherent.cc:266).
// Here time advances in blocks (one block = L / fs seconds), so the loop is deterministic: call
// on_block() once per ccoherent::step().  A free-running thread like the reference's threadf needs
// real dongles; the maths (descent, hold time, threshold) is the reference's.
#ifndef CCONTROLH
#define CCONTROLH
#include <cmath>
#include "common.h"

class csyntheticsdr;

class ccontrol {
    csyntheticsdr *dev;
    int hold_blocks;      // blocks left at the altered sample rate
    bool correcting;
public:
    static constexpr double maxppm = 8192.0 / 16777216.0; // TWO_POW(13)/TWO_POW(24) = 2^-11, src/ccontrol.cc:27
    static constexpr float scale = 100.0f;                // :28
    static constexpr float frac_t = 0.90f;                // :29
    static float descent(float lag) { return (float)(maxppm * std::tanh(lag / scale)); } // :73-76
    // the sample rate the RTL2832 really runs at for a requested one (realfs(), :36-45: 28.8 MHz crystal, 2^22 / ratio with
    // the two low bits of the ratio masked off)
    static double realfs(uint32_t requestedfs)
    {
        const double xtal22 = 28800000.0 * 4194304.0;
        uint32_t fsratio = (uint32_t)(xtal22 / requestedfs);
        fsratio &= 0x0ffffffcu;
        const uint32_t real_fsratio = fsratio | ((fsratio & 0x08000000u) << 1);
        return xtal22 / real_fsratio;
    }
    static bool needs_correction(float lag) { return std::fabs(lag) > sync_threshold; }   // :99
    // time to spend at the altered sample rate, :101-102 with the reference's types (p, fs and the quotient are float)
    static double hold_seconds(float lag, uint32_t samplerate)
    {
        const float fs = (float)realfs(samplerate);
        const float p = descent(lag);
        return frac_t * std::fabs(lag / (p * fs));
    }
    // the model's clock is the block: the nanosleep of :110 becomes ceil(t / (L / fs)) whole blocks, at least one
    static int hold_block_count(float lag, uint32_t samplerate, uint32_t blocksize)
    {
        const double block_seconds = (double)(blocksize >> 1) / (double)samplerate;
        const int n = (int)std::ceil(hold_seconds(lag, samplerate) / block_seconds);
        return n < 1 ? 1 : n;
    }
    explicit ccontrol(csyntheticsdr *d) : dev(d), hold_blocks(0), correcting(false) {}
    void on_block();      // one iteration of threadf's loop body (:91-119) in block time
    bool is_correcting() const { return correcting; }
};
#endif
