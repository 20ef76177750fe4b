// Test tool (std headers only; compiled by tests/test_egs.py): the reference's minibatch order, stated the way
// /root/reference/src/my_lib_example_rand.cpp:41,69-93,119-141 states it -- the containers, the iteration and the
// std::shuffle calls are the reference's, so what this prints IS the order a reference built against the same standard
// library forms -- for io.RandExample(order="reference") to be compared with.
//   reference_batch_order LEN_FILE SEED BATCHSIZE EPOCHS  ->  one line per batch "epoch: key key ...", epochs in order
#include <algorithm>
#include <fstream>
#include <iostream>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

int main(int argc, char **argv) {
  if (argc != 5) return 2;
  const int seed = std::stoi(argv[2]);
  const size_t batchsize = (size_t)std::stoi(argv[3]);
  const int epochs = std::stoi(argv[4]);
  std::unordered_map<size_t, std::vector<std::string>> length_to_keys;
  {  // "key length" pairs, token by token (my_lib_example_rand.cpp:79-93)
    std::ifstream in(argv[1]);
    if (!in.is_open()) return 3;
    std::string tok, key;
    bool is_key = true;
    while (in) {
      in >> tok;
      if (is_key)
        key = tok;
      else
        length_to_keys[std::stoi(tok)].push_back(key);
      is_key = !is_key;
    }
  }
  std::mt19937 engine(seed);
  for (int epoch = 0; epoch < epochs; ++epoch) {  // the constructor's shuffle_keys(), then one per reset()
    std::vector<std::vector<std::string>> key_batch;
    std::vector<std::string> batch;
    for (auto kv : length_to_keys) {  // (a copy, as there: the map's vectors keep the file order)
      auto &keys = kv.second;
      std::shuffle(keys.begin(), keys.end(), engine);
      for (const auto &k : keys) {
        batch.push_back(k);
        if (batch.size() == batchsize) {
          key_batch.push_back(batch);
          batch.clear();
        }
      }
      if (batch.size() > 0) key_batch.push_back(batch);
      batch.clear();
    }
    std::shuffle(key_batch.begin(), key_batch.end(), engine);
    for (const auto &b : key_batch) {
      std::cout << epoch << ":";
      for (const auto &k : b) std::cout << " " << k;
      std::cout << "\n";
    }
  }
  return 0;
}
