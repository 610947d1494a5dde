/*! \file  FRIES/hh_vec.hpp for the MI355X build: HubHolVec<el_type>, the DistVec of Hubbard-Holstein basis states, with the reference's
 * public members (FRIES/hh_vec.hpp:11-262).  Index = [alpha sites | beta sites | ph_bits bits per site] (at most 64 bits here:
 * n_sites <= 12 with 3 phonon bits).  Per stored state the vector keeps the lists of electrons that can hop right / left (neighb()) and
 * the phonon numbers of the sites (phonon_nums()); for the device-bound vector both are host mirrors re-derived from the determinant
 * mirror whenever that is refreshed.  The vector moves to the device at the first comp_sub of the driver's loop
 * (FRIES/compress_utils.hpp of this build), with the context set up by fries_hh_setup from the parameters parse_hh_input read. */
#ifndef hh_vec_h
#define hh_vec_h
#include <FRIES/vec_utils.hpp>


template <class el_type>
class HubHolVec : public DistVec<el_type> {
    typedef DistVec<el_type> Base;
    Matrix<uint8_t> neighb_;
    uint8_t n_sites_, ph_bits_;
    Matrix<uint8_t> phonon_nums_;
    std::vector<uint32_t> vec_rns_, proc_rns_hh_;
    uint64_t word_of(const uint8_t *det) const { uint64_t w = 0; memcpy(&w, det, Base::indices_.cols() > 8 ? 8 : Base::indices_.cols()); return w; }
    uint64_t elec_part(const uint8_t *det) const { return word_of(det) & ((1ull << (2 * n_sites_)) - 1ull); }
protected:
    void mirror_refreshed(size_t n) override {
        for (size_t p = 0; p < n; p++) { find_neighbors_1D(Base::indices_[p], neighb_[p]); decode_phonons(Base::indices_[p], phonon_nums_[p]); gen_orb_list(Base::indices_[p], Base::occ_orbs_[p]); }
    }
    void device_setup(fries_ctx *cx, uint32_t /*mat_nonz*/, bool /*new_hb*/) override {
        const fries_hip::HHParams &P = fries_hip::hh_params();
        if (!P.set) throw std::runtime_error("parse_hh_input must run before the Hubbard-Holstein vector moves to the device");
        fries_hip::ck(fries_set_proc_scrambler(cx, proc_rns_hh_.data(), proc_rns_hh_.size()));
        fries_hip::ck(fries_set_vec_scrambler(cx, vec_rns_.data(), vec_rns_.size()));
        fries_hh_params hp{};
        hp.n_elec = P.n_elec; hp.n_sites = P.lat_len; hp.eps = P.eps; hp.U = P.U; hp.omega = P.omega; hp.g = P.g; hp.gs_energy = P.gs_energy;
        hp.vec_nonz = hh_vec_nonz; hp.max_dets = (uint32_t)Base::max_size_;
        fries_hip::ck(fries_hh_setup(cx, &hp));
    }
public:
    uint32_t hh_vec_nonz = 0;          // (MI355X build) the sample budget of the driver's comp_sub calls, known at the first one
    bool hh_candidate() const override { return !Base::bound() && Base::num_vecs() == 2; }
    void hh_budget(uint32_t n_samp) override { hh_vec_nonz = n_samp; }
    HubHolVec(size_t size, size_t add_size, uint8_t n_sites, uint8_t max_ph, unsigned int n_elec, int n_procs, std::function<double(const uint8_t *)> diag_fxn,
              uint8_t n_vecs, std::vector<uint32_t> rns_common, std::vector<uint32_t> rns_distinct) :
        DistVec<el_type>(size, add_size, (uint8_t)(n_sites * 2 + n_sites * max_ph), n_elec, n_procs, diag_fxn, n_vecs, rns_common, rns_distinct),
        neighb_(size, 2 * (n_elec + 1)), n_sites_(n_sites), ph_bits_(max_ph), phonon_nums_(size, n_sites), vec_rns_(rns_distinct), proc_rns_hh_(rns_common) {
        if (n_sites * (2 + max_ph) > 64) throw std::runtime_error("this build stores a Hubbard-Holstein state in 64 bits");
        fries_hip::Backend::get().hh_mode = true;
    }
    /* the electrons only (hh_vec.hpp:41-52) */
    uint8_t gen_orb_list(uint8_t *det, uint8_t *occ_orbs) override {
        uint8_t n = 0;
        for (uint64_t e = elec_part(det); e; e &= e - 1) occ_orbs[n++] = (uint8_t)__builtin_ctzll(e);
        return n;
    }
    /* hashes of the occupied orbitals followed by every site's phonon number (hh_vec.hpp:54-88) */
    int idx_to_proc(uint8_t *idx) override {
        uint8_t orbs[64], ph[64];
        const uint8_t ne = gen_orb_list(idx, orbs);
        decode_phonons(idx, ph);
        int n_procs = 1;
        MPI_Comm_size(MPI_COMM_WORLD, &n_procs);
        return (int)(Base::proc_hash_.hash_fxn(orbs, ne, ph, n_sites_) % (uintmax_t)n_procs);
    }
    int idx_to_proc(uint8_t *idx, uint8_t * /*orbs*/) override { return idx_to_proc(idx); }
    uintmax_t idx_to_hash(uint8_t *idx, uint8_t *orbs) override {
        const unsigned int want = (unsigned int)Base::occ_orbs_.cols();
        if (gen_orb_list(idx, orbs) != want) throw std::runtime_error("Determinant created with an incorrect number of electrons");
        uint8_t ph[64];
        decode_phonons(idx, ph);
        return Base::vec_hash_.hash_fxn(orbs, (uint8_t)want, ph, n_sites_);
    }
    void expand() override {
        Base::expand();
        neighb_.reshape(Base::max_size_, neighb_.cols());
        phonon_nums_.reshape(Base::max_size_, phonon_nums_.cols());
    }
    Matrix<uint8_t> &neighb() { Base::indices(); return neighb_; }
    Matrix<uint8_t> &phonon_nums() { Base::indices(); return phonon_nums_; }
    uint8_t *phonons_at_pos(size_t pos) { Base::indices(); return phonon_nums_[pos]; }
    uint8_t tot_ph_at_idx(size_t idx) { return (uint8_t)total_ph(idx); }
    /* electrons whose right / left neighbour site (same spin) is empty, open ends (hh_vec.hpp:139-175) */
    void find_neighbors_1D(uint8_t *det, uint8_t *neighbors) {
        const unsigned int L = n_sites_, n_elec = (unsigned int)Base::occ_orbs_.cols();
        const uint64_t e = elec_part(det);
        uint64_t right = e & ~(e >> 1);                     // occupied, next orbital up empty
        right &= ~(1ull << (L - 1)) & ~(1ull << (2 * L - 1));       // the last site of either spin has no right neighbour
        uint64_t left = e & (~e << 1);                      // occupied, next orbital down empty
        left &= ~(1ull << L) & ~1ull;                       // the first site of either spin has no left neighbour
        uint8_t n = 0;
        for (uint64_t r = right; r; r &= r - 1) neighbors[1 + n++] = (uint8_t)__builtin_ctzll(r);
        neighbors[0] = n;
        n = 0;
        for (uint64_t l = left; l; l &= l - 1) neighbors[n_elec + 2 + n++] = (uint8_t)__builtin_ctzll(l);
        neighbors[n_elec + 1] = n;
    }
    void decode_phonons(uint8_t *det, uint8_t *numbers) {
        const uint64_t w = word_of(det) >> (2 * n_sites_);
        for (uint8_t s = 0; s < n_sites_; s++) numbers[s] = (uint8_t)((w >> (s * ph_bits_)) & ((1u << ph_bits_) - 1u));
    }
    /* the state with one phonon more / less on a site; 0 when the number would leave [0, 2^ph_bits) (hh_vec.hpp:196-224) */
    int det_from_ph(uint8_t *orig, uint8_t *new_det, uint8_t site_idx, int change) {
        const unsigned int at = 2 * n_sites_ + site_idx * ph_bits_;
        uint64_t w = word_of(orig);
        const unsigned int top = (1u << ph_bits_) - 1u, now = (unsigned int)((w >> at) & top);
        if (change == 1 && now == top) { std::cerr << "Warning: maximum phonon number reached\n"; return 0; }
        if (change == -1 && now == 0) return 0;
        w = (w & ~((uint64_t)top << at)) | ((uint64_t)(now + change) << at);
        memcpy(new_det, &w, CEILING(Base::n_bits_, 8));
        return 1;
    }
    unsigned int total_ph(size_t idx) {
        Base::indices();
        unsigned int s = 0;
        for (uint8_t k = 0; k < n_sites_; k++) s += phonon_nums_(idx, k);
        return s;
    }
    void initialize_at_pos(size_t pos, uint8_t *orbs) override {
        Base::initialize_at_pos(pos, orbs);
        find_neighbors_1D(Base::indices_[pos], neighb_[pos]);
        decode_phonons(Base::indices_[pos], phonon_nums_[pos]);
    }
    /* the diagonal function sees the whole index, not the occupied-orbital list (hh_vec.hpp:253-259) */
    double matr_el_at_pos(size_t pos) {
        Base::indices();
        return Base::diag_calc_(Base::indices_[pos]);
    }
};
#endif /* hh_vec_h */
