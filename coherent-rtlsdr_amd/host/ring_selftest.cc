// ring_selftest.cc -- the block ring under overrun, no GPU: a free-running producer thread (csyntheticsdr::start, the
// librtlsdr callback thread of src/crtlsdr.cc:61-68,173-193) against a consumer that is slower than it and copies every
// block OUTSIDE the lock, like ccoherent::step does.  Every block the consumer gets must be whole (equal to the
// generator's row for the readcnt it carries -- a slot reused under the reader would tear it), readcnts must rise
// strictly, and blocks must really have been dropped.  Prints "RING OK <consumed> <overruns>".
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "csdrdevice.h"

int main(int argc, char **argv)
{
    const int L = 512, nblocks = argc > 1 ? atoi(argv[1]) : 400;
    const uint32_t B = 2 * L;
    csynthsource src(2, L, 777, 16, false);
    csyntheticsdr dev(&src, 1, B);
    dev.start(/*pace_us*/ 50, /*max_blocks*/ nblocks);
    std::vector<int8_t> copy(B), expect(B);
    long consumed = 0, torn = 0, order = 0;
    uint32_t last = 0;
    bool first = true;
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::seconds(20);
    while (std::chrono::steady_clock::now() < t_end && !dev.drained((uint32_t)nblocks)) {
        int8_t *p = dev.read();
        const uint32_t rc = dev.get_readcntbuf();
        // slow copy in two halves with a pause between them: a producer that reused this slot would change the second half
        std::memcpy(copy.data(), p, B / 2);
        std::this_thread::sleep_for(std::chrono::microseconds(400));
        std::memcpy(copy.data() + B / 2, p + B / 2, B / 2);
        dev.consume();
        csynth_make_row(src.get_params(), (int)rc, 1, -1.0, expect.data());
        for (uint32_t i = 0; i < B; ++i) expect[i] = (int8_t)((uint8_t)expect[i] ^ 0x80u);     // the ring holds raw offset binary
        if (std::memcmp(copy.data(), expect.data(), B) != 0) ++torn;
        if (!first && rc <= last) ++order;
        first = false; last = rc;
        ++consumed;
    }
    dev.stop();
    const uint32_t over = dev.get_overruns();
    std::printf("consumed %ld of %d blocks, %u overruns, %ld torn, %ld out of order\n", consumed, nblocks, over, torn, order);
    if (torn || order || over == 0 || consumed + (long)over > nblocks || consumed < 8) return 1;
    std::printf("RING OK %ld %u\n", consumed, over);
    return 0;
}
