// packetizer_selftest.cc -- the wire format without a GPU: synthetic rows -> cpacketize::write(int8*) ->
// notifysend -> send (ZMQ PUB + debug PUB).  tests/test_wire_format.py subscribes and parses the packets
// the way matlabclient/zmqsdr.c:118-144 does.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <unistd.h>
#include <vector>
#include "cpacketizer.h"
extern "C" {
#include "csynth.h"
}
int main(int argc, char **argv)
{
    int nsig = 3, L = 256, blocks = 40, pace_ms = 20;
    std::string addr = "tcp://127.0.0.1:5555";
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string a = argv[i];
        if (a == "--nsig") nsig = atoi(argv[i + 1]);
        else if (a == "--L") L = atoi(argv[i + 1]);
        else if (a == "--blocks") blocks = atoi(argv[i + 1]);
        else if (a == "--pace-ms") pace_ms = atoi(argv[i + 1]);
        else if (a == "--zmq") addr = argv[i + 1];
        else if (a == "--zmq-debug") cpacketize::debugaddress = argv[i + 1];
        else if (a == "--refpadding") cpacketize::refpadding = atoi(argv[i + 1]) != 0;   // the reference's 2x data length on the wire
    }
    const uint32_t B = 2 * (uint32_t)L, N = 1 + (uint32_t)nsig;
    std::vector<cpacketize> channels(N);               // one cpacketize member per device (src/csdrdevice.cc:20-48)
    cpacketize::init(addr, false, N, B);
    if (!cpacketize::publishing()) { std::fprintf(stderr, "zmq unavailable\n"); return 3; }
    csynth_params *p = csynth_params_create(nsig, L, 4242, L / 8, 0);
    std::vector<int8_t> rows((size_t)N * B);
    usleep(300 * 1000);                                 // let subscribers join (PUB/SUB slow joiner)
    for (int t = 0; t < blocks; ++t) {
        csynth_make_block(p, t, -1.0, rows.data());
        for (uint32_t c = 0; c < N; ++c) {
            channels[c].write(c, 1000 + t, rows.data() + (size_t)c * B);
            channels[c].writedebug(c, std::complex<float>((float)c, (float)t));
        }
        channels[0].notifysend();
        cpacketize::send();
        usleep(pace_ms * 1000);
    }
    csynth_params_destroy(p);
    cpacketize::cleanup();
    return 0;
}
