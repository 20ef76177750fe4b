#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "torchain_hip.h"
// Reads every prefix-truncated and byte-flipped variant of an example file through tc_example_read (ASan/UBSan build).
int main(int argc, char **argv) {
  const char *path = argv[1];
  FILE *f = fopen(path, "rb");
  std::vector<unsigned char> data;
  int c;
  while ((c = fgetc(f)) != EOF) data.push_back((unsigned char)c);
  fclose(f);
  const char *tmp = "/tmp/asan_egs/variant.bin";
  int ok = 0, bad = 0;
  auto run = [&](const std::vector<unsigned char> &v) {
    FILE *g = fopen(tmp, "wb");
    fwrite(v.data(), 1, v.size(), g);
    fclose(g);
    const char *paths[2] = {tmp, tmp};
    int64_t offs[2] = {0, 0};
    tc_example *ex = nullptr;
    int rc = tc_example_read(paths, offs, 2, 1, &ex);
    if (rc == 0) { ++ok; tc_example_free(ex); } else ++bad;
  };
  run(data);
  for (size_t n = 0; n < data.size(); n += (data.size() > 4000 ? 97 : 1)) run(std::vector<unsigned char>(data.begin(), data.begin() + n));
  srand(1);
  for (int i = 0; i < 3000; ++i) {
    std::vector<unsigned char> v = data;
    const size_t pos = (size_t)rand() % v.size();
    v[pos] ^= (unsigned char)(1 + rand() % 255);
    run(v);
  }
  printf("ok %d refused %d\n", ok, bad);
  return 0;
}
