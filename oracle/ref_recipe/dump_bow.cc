// bow_vec / bow_feat_vec of the REFERENCE's compute_bow (data/bow_vocabulary.cc:18-24) for the descriptors of the ORB fixtures, with the real
// FBoW (default build) or DBoW2 (-DUSE_DBOW2) and a real vocabulary file: pins stella_vslam_amd/data.py bow_vocabulary / svgpu_fbow_transform,
// which restate FBoW from its published sources (parity unpinned until these fixtures exist).  Compiles only where FBoW / DBoW2 are installed.
//   oracle/_ref/build/dump_bow <orb_vocab.fbow> oracle/_ref/fixtures      (reads ref_orb_640x480_f0_desc.npy written by dump_fixtures)
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include <opencv2/core.hpp>

#include "stella_vslam/data/bow_vocabulary.h"

namespace {
void write_npy(const std::string& path, const char* descr, const std::vector<size_t>& shape, const void* data, size_t bytes) {
    std::string hdr = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': (";
    for (size_t i = 0; i < shape.size(); ++i) hdr += std::to_string(shape[i]) + (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
    hdr += "), }";
    while ((10 + hdr.size() + 1) % 64) hdr += ' ';
    hdr += '\n';
    std::ofstream f(path, std::ios::binary);
    const char magic[] = "\x93NUMPY\x01\x00";
    f.write(magic, 8);
    const uint16_t n = (uint16_t)hdr.size();
    f.write((const char*)&n, 2);
    f.write(hdr.data(), hdr.size());
    f.write((const char*)data, bytes);
}
cv::Mat read_desc_npy(const std::string& path) {  // the |u1 (n, 32) arrays dump_fixtures writes
    std::ifstream f(path, std::ios::binary);
    char magic[8];
    f.read(magic, 8);
    uint16_t hl = 0;
    f.read((char*)&hl, 2);
    std::string hdr(hl, ' ');
    f.read(&hdr[0], hl);
    const size_t a = hdr.find("(") + 1;
    const int n = std::stoi(hdr.substr(a));
    cv::Mat d(n, 32, CV_8UC1);
    f.read((char*)d.data, (std::streamsize)n * 32);
    return d;
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: dump_bow <vocabulary file> <fixtures dir>\n");
        return 2;
    }
    using namespace stella_vslam;
    const std::string dir = argv[2];
    auto* vocab = data::bow_vocabulary_util::load(argv[1]);
    const cv::Mat desc = read_desc_npy(dir + "/ref_orb_640x480_f0_desc.npy");
    data::bow_vector bow_vec;
    data::bow_feature_vector bow_feat_vec;
    data::bow_vocabulary_util::compute_bow(vocab, desc, bow_vec, bow_feat_vec);
    std::vector<double> words;  // (word id, weight) rows
    for (const auto& kv : bow_vec) {
        words.push_back((double)kv.first);
        words.push_back((double)kv.second);
    }
    std::vector<int64_t> feats;  // (node key, feature index) rows, in map / push_back order
    for (const auto& kv : bow_feat_vec)
        for (const auto idx : kv.second) {
            feats.push_back((int64_t)kv.first);
            feats.push_back((int64_t)idx);
        }
    write_npy(dir + "/ref_bow_vec.npy", "<f8", {words.size() / 2, 2}, words.data(), words.size() * 8);
    write_npy(dir + "/ref_bow_feat_vec.npy", "<i8", {feats.size() / 2, 2}, feats.data(), feats.size() * 8);
    std::printf("%zu words, %zu features in %zu nodes\n", bow_vec.size(), feats.size() / 2, bow_feat_vec.size());
    return 0;
}
