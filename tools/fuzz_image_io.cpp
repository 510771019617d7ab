// fuzz_image_io — mutation fuzzing of the texture readers (image_io.cpp, jpeg_io.cpp) under AddressSanitizer/UBSan on the CPU build:
// every input must end in a decoded image or in a std::exception, never in a crash, an out-of-bounds access or a hang.
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined tools/fuzz_image_io.cpp \
//       magr_ray_tracer_amd/host/{image_io,jpeg_io,scene_build,scene_io,accel_build}.cpp -lz -pthread -o /tmp/fuzz_image_io
//   /tmp/fuzz_image_io <iterations> seed1.png seed2.jpg ...
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <fstream>
#include <random>
#include <string>
#include <vector>
#include "../magr_ray_tracer_amd/host/rt_host.h"

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: fuzz_image_io <iterations> <seed files...>\n"); return 2; }
    const long iters = atol(argv[1]);
    std::mt19937 rng(argc > 100 ? 1 : (unsigned)std::chrono::steady_clock::now().time_since_epoch().count());
    long ok = 0, rejected = 0;
    for (int a = 2; a < argc; a++) {
        std::ifstream in(argv[a], std::ios::binary);
        std::vector<char> seed((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (seed.empty()) { fprintf(stderr, "cannot read %s\n", argv[a]); return 2; }
        const std::string name = argv[a];
        const std::string ext = name.substr(name.find_last_of('.'));
        const std::string tmp = "/tmp/fuzz_case" + ext;
        for (long it = 0; it < iters; it++) {
            std::vector<char> b = seed;
            const int kind = (int)(rng() % 5);
            const int n = 1 + (int)(rng() % 8);
            for (int k = 0; k < n; k++) {
                const size_t p = rng() % b.size();
                switch (kind) {
                case 0: b[p] = (char)(b[p] ^ (1 << (rng() % 8))); break;                 // bit flip
                case 1: b[p] = (char)rng(); break;                                        // random byte
                case 2: b.resize(std::max<size_t>(1, p)); k = n; break;                   // truncate
                case 3: b.insert(b.begin() + (long)p, (char)rng()); break;                // insert
                default: { const size_t q = rng() % b.size(); b[p] = b[q]; break; }       // copy byte
                }
            }
            { std::ofstream out(tmp, std::ios::binary); out.write(b.data(), (std::streamsize)b.size()); }
            try {
                int w = 0, h = 0;
                const std::vector<float> px = rt355::LoadImageF(tmp, w, h);
                if (px.size() != (size_t)w * h * 3) { fprintf(stderr, "size mismatch on %s iteration %ld\n", name.c_str(), it); return 1; }
                ok++;
            } catch (const std::exception&) { rejected++; }
        }
        printf("%s: done\n", name.c_str());
    }
    printf("fuzz_image_io: %ld decoded, %ld rejected, no crash\n", ok, rejected);
    return 0;
}
