// Sanitizer harness for the product's container reader (flo_amd/csrc/container.cpp): parses every file given on the
// command line and several thousand damaged variants of each (truncations, single- and multi-byte damage driven by a
// fixed LCG). Built with -fsanitize=address,undefined by tests/test_sanitizers.py; any out-of-bounds read or
// undefined operation in the reader aborts the run.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../flo_amd/csrc/container.hpp"

static std::vector<unsigned char> read_all(const char *path) {
    std::vector<unsigned char> v;
    FILE *f = fopen(path, "rb");
    if (!f) return v;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

int main(int argc, char **argv) {
    unsigned long long ok = 0, rejected = 0;
    unsigned int lcg = 12345u;
    auto rnd = [&]() { lcg = lcg * 1103515245u + 12345u; return lcg >> 8; };
    for (int a = 1; a < argc; a++) {
        std::vector<unsigned char> good = read_all(argv[a]);
        if (good.empty()) { fprintf(stderr, "cannot read %s\n", argv[a]); return 2; }
        auto probe = [&](const unsigned char *p, size_t n) {
            // exact-size heap copy so that any read past the end is seen by the sanitizer
            unsigned char *h = (unsigned char *)malloc(n ? n : 1);
            memcpy(h, p, n);
            flo::ParsedFile f;
            const char *err = "";
            int rc = flo::parse_file(h, n, f, &err);
            if (rc == 0) {
                ok++;
                for (const auto &cd : f.channels_desc)
                    if (cd.off + cd.len > n) { fprintf(stderr, "payload outside the file accepted\n"); exit(3); }
            } else {
                rejected++;
                if (!err || !*err) { fprintf(stderr, "rejection without a message\n"); exit(4); }
            }
            free(h);
        };
        probe(good.data(), good.size());
        for (size_t n = 0; n < good.size() && n < 200; n++) probe(good.data(), n);
        for (int i = 0; i < 300; i++) probe(good.data(), rnd() % good.size());
        std::vector<unsigned char> bad;
        for (int i = 0; i < 3000; i++) {
            bad = good;
            int k = 1 + (int)(rnd() % 4);
            for (int j = 0; j < k; j++) {
                size_t pos = (i % 3 == 0) ? rnd() % (good.size() < 200 ? good.size() : 200) : rnd() % good.size();
                bad[pos] = (unsigned char)rnd();
            }
            probe(bad.data(), bad.size());
        }
    }
    printf("accepted %llu rejected %llu\n", ok, rejected);
    return (ok > 0 && rejected > 0) ? 0 : 5;
}
