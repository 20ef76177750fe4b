#include <cstdio>
#include <cstdlib>
#include <vector>
#include "torchain_hip.h"
// tc_den_graph_read over truncated and byte-flipped variants of a den.fst (ASan/UBSan build, host only).
int main(int argc, char **argv) {
  FILE *f = fopen(argv[1], "rb");
  std::vector<unsigned char> data;
  int c;
  while ((c = fgetc(f)) != EOF) data.push_back((unsigned char)c);
  fclose(f);
  const int P = atoi(argv[2]);
  const char *tmp = "/tmp/asan_fst/variant.fst";
  int ok = 0, bad = 0;
  auto run = [&](const std::vector<unsigned char> &v) {
    FILE *g = fopen(tmp, "wb");
    if (!v.empty()) fwrite(v.data(), 1, v.size(), g);
    fclose(g);
    tc_den_graph *h = nullptr;
    int rc = tc_den_graph_read(&h, tmp, P);
    if (rc == 0) { ++ok; tc_den_graph_free(h); } else ++bad;
  };
  run(data);
  // the accepted variants also through the streamed path's list builder, both slab widths, tied and general
  const char *modes[][3] = {{"force_streamed", "slab_narrow", nullptr}, {"force_streamed", "slab_wide", nullptr},
                            {"force_streamed", "force_general", "slab_wide"}};
  for (auto &m : modes) {
    for (const char *k : m) if (k) tc_debug_set(k, 1);
    srand(7);
    for (int i = 0; i < 300; ++i) {
      std::vector<unsigned char> v = data;
      const size_t pos = (size_t)rand() % v.size();
      v[pos] ^= (unsigned char)(1 + rand() % 255);
      run(v);
    }
    for (const char *k : m) if (k) tc_debug_set(k, 0);
  }
  for (size_t n = 0; n < data.size(); n += (data.size() > 3000 ? 53 : 1)) run(std::vector<unsigned char>(data.begin(), data.begin() + n));
  srand(2);
  for (int i = 0; i < 1500; ++i) {
    std::vector<unsigned char> v = data;
    const size_t pos = (size_t)rand() % v.size();
    v[pos] ^= (unsigned char)(1 + rand() % 255);
    run(v);
  }
  printf("ok %d refused %d\n", ok, bad);
  return 0;
}
