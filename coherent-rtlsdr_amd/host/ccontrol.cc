// ccontrol.cc -- see ccontrol.h.
#include "ccontrol.h"
#include "csdrdevice.h"

void ccontrol::on_block()
{
    if (correcting) {                                   // nanosleep(t) at the altered rate, src/ccontrol.cc:110
        dev->advance_resampler();
        if (--hold_blocks <= 0) {
            dev->set_correction_f(0.0f);                // :113
            correcting = false;
            dev->requestfft();                          // next loop iteration: requestfftblocking() :91
        }
        return;
    }
    if (dev->get_synchronized()) return;                // wait_synchronized() :89
    if (dev->is_lagrequested()) return;                 // the requested lag has not arrived yet
    const float lag = dev->get_lagp()->lag;             // :93
    const float fs = dev->get_samplerate();
    const double block_seconds = (double)(dev->get_blocksize() >> 1) / fs;
    if (std::fabs(lag) > sync_threshold) {              // :99
        const float p = descent(lag);                   // :101
        const double t = frac_t * std::fabs(lag / (p * fs)); // :102 time to spend at the altered sample rate
        dev->set_correction_f(p);                       // :108
        hold_blocks = (int)std::ceil(t / block_seconds);
        if (hold_blocks < 1) hold_blocks = 1;
        correcting = true;
    } else {
        dev->set_correction_f(0.0f);                    // :116-117
        dev->set_synchronized(true);
    }
}
