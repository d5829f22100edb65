// nequip_amd_aoti_run <package.nequip.pt2> <dir> -- evaluate a compiled model without a Python interpreter.
//
// What a C++ host of the reference's compiled models does (the LAMMPS pair style of pair_nequip_allegro: load the
// AOTInductor package, make the custom-op libraries it names known to the dispatcher, hand over the input tensors in the
// order of the package's `nequip_aoti_inputs` metadata, read the outputs in the order of `nequip_aoti_outputs`;
// nequip/model/inference_models/aotinductor.py:57-125 is the Python form of the same steps).  Here the op library is
// libnequip_amd_torch.so, linked directly: its TORCH_LIBRARY block runs when this executable is loaded.
//
// <dir>/inputs.txt: one line per input, `file dtype ndim dims...` (dtype f32 | f64 | i64, raw little-endian data in
// <dir>/file).  Writes <dir>/out<i>.bin and <dir>/outputs.txt in the same format.
#include <ATen/ATen.h>
#include <torch/csrc/inductor/aoti_package/model_package_loader.h>

#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "nequip_amd_torch.h"

static at::ScalarType dtype_of(const std::string& s) {
  if (s == "f32") return at::kFloat;
  if (s == "f64") return at::kDouble;
  if (s == "i64") return at::kLong;
  throw std::runtime_error("unknown dtype " + s);
}

static const char* name_of(at::ScalarType t) {
  switch (t) {
    case at::kFloat: return "f32";
    case at::kDouble: return "f64";
    case at::kLong: return "i64";
    default: throw std::runtime_error("unsupported output dtype");
  }
}

int main(int argc, char** argv) {
  if (argc != 3) {
    std::fprintf(stderr, "usage: %s <package.nequip.pt2> <dir with inputs.txt>\n", argv[0]);
    return 2;
  }
  try {
    if (nqa_torch_ops_registered_here() != 1) return 3;  // (keeps the op library linked)
    const std::string pkg = argv[1], dir = argv[2];
    std::vector<at::Tensor> inputs;
    std::ifstream manifest(dir + "/inputs.txt");
    if (!manifest) throw std::runtime_error("cannot open " + dir + "/inputs.txt");
    std::string line;
    while (std::getline(manifest, line)) {
      if (line.empty()) continue;
      std::istringstream is(line);
      std::string file, dt;
      int nd = 0;
      is >> file >> dt >> nd;
      std::vector<int64_t> shape((size_t)nd);
      for (auto& s : shape) is >> s;
      at::Tensor t = at::empty(shape, at::TensorOptions().dtype(dtype_of(dt)));
      std::ifstream f(dir + "/" + file, std::ios::binary);
      if (!f) throw std::runtime_error("cannot open " + dir + "/" + file);
      f.read(static_cast<char*>(t.data_ptr()), (std::streamsize)t.nbytes());
      if ((size_t)f.gcount() != t.nbytes()) throw std::runtime_error("short read of " + file);
      inputs.push_back(t.to(at::kCUDA));
    }
    torch::inductor::AOTIModelPackageLoader loader(pkg);
    std::vector<at::Tensor> outputs = loader.run(inputs);
    std::ofstream out_manifest(dir + "/outputs.txt");
    for (size_t i = 0; i < outputs.size(); ++i) {
      const at::Tensor t = outputs[i].cpu().contiguous();
      const std::string file = "out" + std::to_string(i) + ".bin";
      std::ofstream f(dir + "/" + file, std::ios::binary);
      f.write(static_cast<const char*>(t.data_ptr()), (std::streamsize)t.nbytes());
      out_manifest << file << " " << name_of(t.scalar_type()) << " " << t.dim();
      for (int64_t s : t.sizes()) out_manifest << " " << s;
      out_manifest << "\n";
    }
    std::printf("[nequip_amd_aoti_run] %zu inputs -> %zu outputs\n", inputs.size(), outputs.size());
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "[nequip_amd_aoti_run] %s\n", e.what());
    return 1;
  }
}
