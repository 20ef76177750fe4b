// The measured kernel choices of earlier runs (api.cpp: tune_den_variant), kept by the LIBRARY: a C-side integrator of the
// reference's denominator-graph handle (src/my_lib.h:29) that calls tc_den_graph_prepare directly gets the same kernel --
// and with it the same last bits -- from run to run without knowing about tc_den_graph_set_variant.  One JSON file,
// $TORCHAIN_TUNING_CACHE or ~/.cache/torchain_amd/tuning.json (read at every call: tests point it elsewhere), an object of
//   "<tc_den_graph_hash as 16 hex digits>:<device name>:k<kernel generation>": {"fused_ms": f, "two_sequence_kernel": 0 | 1, "two_sequence_ms": f}
// entries -- the format torchain_amd/io.py wrote in round 4, so both sides read each other's files.  Any failure (no
// home, read-only directory, damaged file) means "not cached": the choice is timed again.  Host code only.
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

#include "chain_internal.h"

namespace tc {

namespace {

std::string cache_path() {
  const char *env = getenv("TORCHAIN_TUNING_CACHE");
  if (env && *env) return env;
  const char *home = getenv("HOME");
  if (!home || !*home) return "";
  return std::string(home) + "/.cache/torchain_amd/tuning.json";
}

// {"key": {flat body}, ...} -> key -> body text (without the braces).  Tolerant: stops at the first thing it does not expect.
std::map<std::string, std::string> read_table(const std::string &path) {
  std::map<std::string, std::string> table;
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return table;
  std::string text;
  char buf[4096];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) {
    text.append(buf, n);
    if (text.size() > ((size_t)16 << 20)) break;  // (not a tuning cache)
  }
  fclose(f);
  size_t i = text.find('{');
  if (i == std::string::npos) return table;
  ++i;
  for (;;) {
    const size_t k0 = text.find('"', i);
    if (k0 == std::string::npos) break;
    size_t k1 = k0 + 1;
    while (k1 < text.size() && text[k1] != '"') k1 += text[k1] == '\\' ? 2 : 1;
    if (k1 >= text.size()) break;
    const size_t b0 = text.find('{', k1);
    if (b0 == std::string::npos) break;
    const size_t b1 = text.find('}', b0);
    if (b1 == std::string::npos) break;
    // (entries are plain ASCII as this library and the Python writer form them; anything else is damage, and carried into
    // the next file it would make that one unreadable to a JSON parser)
    const std::string key = text.substr(k0 + 1, k1 - k0 - 1), body = text.substr(b0 + 1, b1 - b0 - 1);
    bool plain = true;
    for (unsigned char c : key) plain = plain && c >= 0x20 && c < 0x7f && c != '\\';
    for (unsigned char c : body) plain = plain && ((c >= 0x20 && c < 0x7f && c != '\\' && c != '"') || c == '\n' || c == '\r' || c == '\t' || c == '"');
    if (plain) table[key] = body;
    i = b1 + 1;
  }
  return table;
}

bool field(const std::string &body, const char *name, double *out) {
  const std::string quoted = std::string("\"") + name + "\"";
  const size_t at = body.find(quoted);
  if (at == std::string::npos) return false;
  const size_t colon = body.find(':', at + quoted.size());
  if (colon == std::string::npos) return false;
  char *end = nullptr;
  const double v = strtod(body.c_str() + colon + 1, &end);
  if (end == body.c_str() + colon + 1) return false;
  *out = v;
  return true;
}

void make_dirs(const std::string &path) {
  for (size_t i = 1; i < path.size(); ++i)
    if (path[i] == '/') (void)mkdir(path.substr(0, i).c_str(), 0777);
}

}  // namespace

std::string tuning_cache_key(uint64_t graph_hash, const char *device_name) {
  char hex[24];
  snprintf(hex, sizeof hex, "%016llx", (unsigned long long)graph_hash);
  // (the kernels a choice was timed against: a choice measured with another round's kernels is not this library's)
  std::string key = std::string(hex) + ":" + (device_name ? device_name : "") + ":k" + std::to_string(kKernelGeneration);
  // (the key is written between quotes as it is: nothing in it may need escaping, for this reader or for a JSON parser)
  for (char &c : key)
    if (c == '"' || c == '\\' || (unsigned char)c < 0x20 || (unsigned char)c >= 0x7f) c = '_';
  return key;
}

bool tuning_cache_get(const std::string &key, int *two_sequence_kernel) {
  const std::string path = cache_path();
  if (path.empty()) return false;
  const auto table = read_table(path);
  const auto it = table.find(key);
  double v = 0.0;
  if (it == table.end() || !field(it->second, "two_sequence_kernel", &v)) return false;
  *two_sequence_kernel = v > 0.5 ? 1 : 0;
  return true;
}

void tuning_cache_put(const std::string &key, int two_sequence_kernel, float fused_ms, float two_sequence_ms) {
  const std::string path = cache_path();
  if (path.empty()) return;
  make_dirs(path);
  auto body_of = [](double fused, int choice, double two) {
    char body[160];
    snprintf(body, sizeof body, "\n  \"fused_ms\": %.9g,\n  \"two_sequence_kernel\": %d,\n  \"two_sequence_ms\": %.9g\n ", fused, choice ? 1 : 0, two);
    return std::string(body);
  };
  // entries of the old file are carried over as their three numbers, written anew: whatever else a damaged file held
  // stays behind, and what is written is JSON again
  std::map<std::string, std::string> table;
  for (const auto &kv : read_table(path)) {
    double fused = 0.0, choice = 0.0, two = 0.0;
    if (field(kv.second, "fused_ms", &fused) && field(kv.second, "two_sequence_kernel", &choice) && field(kv.second, "two_sequence_ms", &two) &&
        fused == fused && two == two && fused - fused == 0.0 && two - two == 0.0)
      table[kv.first] = body_of(fused, choice > 0.5, two);
  }
  table[key] = body_of((double)fused_ms, two_sequence_kernel, (double)two_sequence_ms);
  static std::atomic<unsigned> serial{0};  // (two threads of one process preparing graphs at once write two files)
  const std::string tmp = path + "." + std::to_string((long long)getpid()) + "." + std::to_string(serial.fetch_add(1)) + ".tmp";
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return;
  bool ok = fputs("{", f) >= 0;
  bool first = true;
  for (const auto &kv : table) {
    ok = ok && fprintf(f, "%s\n \"%s\": {%s}", first ? "" : ",", kv.first.c_str(), kv.second.c_str()) > 0;
    first = false;
  }
  ok = ok && fputs("\n}", f) >= 0;
  ok = fclose(f) == 0 && ok;
  if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
}

}  // namespace tc

extern "C" {

int tc_tuning_cache_get(uint64_t graph_hash, const char *device_name, int32_t *two_sequence_kernel) {
  if (!device_name || !two_sequence_kernel) return TC_ERR_INVALID_ARGUMENT;
  int v = 0;
  if (!tc::tuning_cache_get(tc::tuning_cache_key(graph_hash, device_name), &v)) return 0;
  *two_sequence_kernel = v;
  return 1;
}

int tc_tuning_cache_put(uint64_t graph_hash, const char *device_name, int32_t two_sequence_kernel, float fused_ms,
                        float two_sequence_ms) {
  if (!device_name || two_sequence_kernel < 0 || two_sequence_kernel > 1) return TC_ERR_INVALID_ARGUMENT;
  tc::tuning_cache_put(tc::tuning_cache_key(graph_hash, device_name), two_sequence_kernel, fused_ms, two_sequence_ms);
  return TC_OK;
}

}  // extern "C"
