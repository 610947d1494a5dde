/*! \file  FRIES/det_hash.hpp for the MI355X build: HashTable<el_type> with the reference's interface (read / del_entry / hash_fxn,
 * FRIES/det_hash.hpp:22-171).  Host-side vectors (trial vectors, a solution vector before it is bound to the device) use it; on the
 * device the table is open-addressed (csrc/vec.hip).  Keys are bit strings of at most 64 bits; the hash value is the reference's:
 * h = 1099511628211 h + (i + 1) * scrambler[occ_i], the product wrapped at 32 bits (:160-170). */
#ifndef det_hash_h
#define det_hash_h
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

template <class el_type>
class HashTable {
    std::unordered_map<uint64_t, el_type> map_;
    std::vector<uint32_t> scrambler_;
    uint8_t idx_size_;
    uint64_t key_of(const uint8_t *idx) const { uint64_t k = 0; memcpy(&k, idx, idx_size_ > 8 ? 8 : idx_size_); return k; }
public:
    HashTable(size_t table_size, std::vector<uint32_t> rn_gen) : scrambler_(rn_gen), idx_size_((uint8_t)((rn_gen.size() + 7) / 8)) { if (table_size) map_.reserve(table_size); }
    /* pointer to the stored value for idx, or NULL if absent and !create; a new entry starts at -1 ("exists, no position yet") */
    el_type *read(uint8_t *idx, uintmax_t /*hash_val*/, bool create) {
        const uint64_t k = key_of(idx);
        auto it = map_.find(k);
        if (it != map_.end()) return &it->second;
        if (!create) return nullptr;
        return &map_.emplace(k, (el_type)-1).first->second;
    }
    void del_entry(uint8_t *idx, uintmax_t /*hash_val*/) { map_.erase(key_of(idx)); }
    uintmax_t hash_fxn(uint8_t *occ_orbs, uint8_t n_elec, uint8_t *phonon_nums, uint8_t n_phonon) {
        uintmax_t hash = 0;
        for (uint8_t i = 0; i < n_elec; i++) hash = 1099511628211ULL * hash + (uint32_t)((i + 1u) * scrambler_[occ_orbs[i]]);
        for (uint8_t i = 0; i < n_phonon; i++) hash = 1099511628211ULL * hash + (uint32_t)((i + 1u) * scrambler_[phonon_nums[i]]);
        return hash;
    }
    size_t size() const { return map_.size(); }
    void print_ht() {}
};
#endif /* det_hash_h */
