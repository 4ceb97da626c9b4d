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
    if (needs_correction(lag)) {                        // :99
        dev->set_correction_f(descent(lag));            // :101,108
        hold_blocks = hold_block_count(lag, (uint32_t)dev->get_samplerate(), dev->get_blocksize());   // :102,110 in block time
        correcting = true;
    } else {
        dev->set_correction_f(0.0f);                    // :116-117
        dev->set_synchronized(true);
    }
}
