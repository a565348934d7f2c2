// A compiled C++ user of include/masp_tx_prover.hpp: what a C++ wallet that held the reference's LocalTxProver would write.
// No Python in its call path: tests/test_tx_prover_cpp.py writes the case file (parameter bytes + descriptions), runs this program
// and compares what it wrote with the Python mirror's results and the oracle's proofs.
//
//   tx_prover_harness --selftest <file>      CPU only: the header's scalar helpers on the vectors of <file>, results as hex lines
//   tx_prover_harness --load <3 paths> 3 x (<bytes> <digest>)   LocalTxProver::from_paths against expected sizes and digests ("mpc": the pinned ones)
//   tx_prover_harness <case.bin> <out.bin>   one MI355X: see the format below
//
// case.bin (little-endian):  "MTP1" | 3 x (u64 length, Parameters bytes: spend, output, convert) | u32 self_verify | u32 threads |
//   u32 batch_cap (0: default) | u32 mode (0: the trait's methods one description at a time, in file order; 1: spend_proofs /
//   output_proofs / convert_proofs over all descriptions of a kind; 2: warm_up, then all Spend descriptions through spend_proofs twice, timed) | u32 n | n records:
//     u32 kind (0 spend, 1 output, 2 convert), then
//     spend:   ak nsk diversifier[11] rcm ar asset value:u64 anchor path[32][32] position:u64 rcv r s
//     output:  esk diversifier[11] pk_d rcm asset value:u64 rcv r s
//     convert: generator value:u64 anchor path[32][32] position:u64 rcv r s
// out.bin: n x (u32 status (1 Some / 0 None = Err(()) / 2 Panic) | zkproof[192] | cv[32] | rk[32]) | bsk[32] | cv_sum[32]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

#include "masp_tx_prover.hpp"

using namespace masp;

struct Reader {
    std::vector<uint8_t> b;
    size_t at = 0;
    void need(size_t n) const {
        if (at + n > b.size()) {
            std::fprintf(stderr, "case file truncated at %zu (+%zu of %zu)\n", at, n, b.size());
            std::exit(2);
        }
    }
    template <class T>
    T num() {
        need(sizeof(T));
        T v = 0;
        for (size_t i = 0; i < sizeof(T); ++i) v |= (T)b[at + i] << (8 * i);
        at += sizeof(T);
        return v;
    }
    template <size_t N>
    std::array<uint8_t, N> bytes() {
        need(N);
        std::array<uint8_t, N> a;
        std::memcpy(a.data(), &b[at], N);
        at += N;
        return a;
    }
    std::vector<uint8_t> blob() {
        const uint64_t n = num<uint64_t>();
        need(n);
        std::vector<uint8_t> v(b.begin() + at, b.begin() + at + n);
        at += n;
        return v;
    }
    MerklePath path() {
        MerklePath p;
        for (auto& node : p.auth_path) node = bytes<32>();
        p.position = num<uint64_t>();
        return p;
    }
};

static std::string hex(const uint8_t* p, size_t n) {
    static const char* d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; ++i) {
        s += d[p[i] >> 4];
        s += d[p[i] & 15];
    }
    return s;
}
static Bytes32 unhex32(const std::string& s) {
    Bytes32 b;
    for (int i = 0; i < 32; ++i) b[i] = (uint8_t)std::stoul(s.substr(2 * i, 2), nullptr, 16);
    return b;
}

// lines "add <a> <b>", "sub <a> <b>", "mul <a> <b>" (jubjub::Fr, 64 hex digits each, little-endian bytes), "pack <32 bytes>", "hstar <a> <b>",
// "sig ..." (a binding signature with a given nonce), "random <count>"
static int selftest(const char* path) {
    std::ifstream f(path);
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream in(line);
        std::string op, a, b;
        in >> op;
        if (op == "add" || op == "sub") {
            in >> a >> b;
            const Bytes32 r = detail::fs_add(unhex32(a), unhex32(b), op == "sub");
            std::printf("%s %s\n", op.c_str(), hex(r.data(), 32).c_str());
        } else if (op == "pack") {
            in >> a;
            uint8_t out[64];
            detail::multipack32(unhex32(a).data(), out);
            std::printf("pack %s %s\n", hex(out, 32).c_str(), hex(out + 32, 32).c_str());
        } else if (op == "hstar") {  // H*(a || b) of RedJubjub: two byte strings of any length as hex ("-" = empty)
            in >> a >> b;
            auto bytes = [](const std::string& x) {
                std::vector<uint8_t> v;
                if (x != "-")
                    for (size_t i = 0; i + 1 < x.size(); i += 2) v.push_back((uint8_t)std::stoul(x.substr(i, 2), nullptr, 16));
                return v;
            };
            const std::vector<uint8_t> va = bytes(a), vb = bytes(b);
            Bytes32 r;
            detail::store(detail::h_star(va.data(), va.size(), vb.data(), vb.size()), r.data());
            std::printf("hstar %s\n", hex(r.data(), 32).c_str());
        } else if (op == "mul") {
            in >> a >> b;
            Bytes32 r;
            detail::store(detail::fs_mul(detail::load(unhex32(a).data()), detail::load(unhex32(b).data())), r.data());
            std::printf("mul %s\n", hex(r.data(), 32).c_str());
        } else if (op == "sig") {  // sig <bsk> <cv_sum> <sighash> <nonce: 80 bytes> <n> n x (<asset identifier> <value: 16 bytes>)
            std::string cv, sh, nonce;
            int n = 0;
            in >> a >> cv >> sh >> nonce >> n;
            std::vector<std::pair<AssetType, I128>> amount;
            for (int i = 0; i < n; ++i) {
                std::string id, val;
                in >> id >> val;
                I128 v;
                for (int k = 0; k < 16; ++k) v.le[k] = (uint8_t)std::stoul(val.substr(2 * k, 2), nullptr, 16);
                amount.push_back({AssetType{unhex32(id)}, v});
            }
            uint8_t t[80];
            for (int k = 0; k < 80; ++k) t[k] = (uint8_t)std::stoul(nonce.substr(2 * k, 2), nullptr, 16);
            const SaplingProvingContext ctx = SaplingProvingContext::from_parts(unhex32(a), unhex32(cv));
            const auto sig = ctx.binding_sig(amount, unhex32(sh).data(), t);
            std::printf("sig %s\n", sig ? hex(sig->data(), 64).c_str() : "None");
        } else if (op == "rcm") {  // Note::rcm of Rseed::AfterZip212(<32 bytes>)
            in >> a;
            const Bytes32 r = Rseed::after_zip212(unhex32(a)).rcm();
            std::printf("rcm %s\n", hex(r.data(), 32).c_str());
        } else if (op == "random") {
            int n = 0, bad = 0, distinct = 1;
            in >> n;
            Bytes32 prev = LocalTxProver::random_scalar();
            for (int i = 1; i < n; ++i) {
                const Bytes32 x = LocalTxProver::random_scalar();
                bad += !detail::fr_canonical(x.data());
                distinct += x != prev;
                prev = x;
            }
            std::printf("random %d canonical %d distinct %d\n", n, n - bad, distinct);
        }
    }
    {   // with_default_location: no folder, no prover (and no panic)
        const char* home = std::getenv("HOME");
        const std::string keep = home ? home : "";
        setenv("HOME", "/nonexistent-masp-home", 1);
        std::printf("default location %s\n", LocalTxProver::with_default_location() ? "found" : "none");
        setenv("HOME", keep.c_str(), 1);
    }
    SaplingProvingContext ctx;  // zero and the identity
    std::printf("context %s %s\n", hex(ctx.bsk().data(), 32).c_str(), hex(ctx.cv_sum().data(), 32).c_str());
    std::printf("selftest ok\n");
    return 0;
}

struct Record {
    uint32_t kind;
    SpendInfo spend;
    OutputInfo output;
    ConvertInfo convert;
    BlindingScalars rs;
    // result
    uint32_t status = 0;
    GrothProofBytes zk{};
    Bytes32 cv{}, rk{};
};

int main(int argc, char** argv) {
    if (argc == 3 && std::string(argv[1]) == "--selftest") return selftest(argv[2]);
    if (argc == 11 && std::string(argv[1]) == "--load") {  // --load <spend> <output> <convert> 3 x (<bytes> <blake2b hex>): LocalTxProver::from_paths
        ExpectedParameterSet e;
        ExpectedParameters* slot[3] = {&e.spend, &e.output, &e.convert};
        for (int k = 0; k < 3; ++k) *slot[k] = ExpectedParameters{(size_t)std::strtoull(argv[5 + 2 * k], nullptr, 10), argv[6 + 2 * k]};
        LocalTxProver::Config cfg;
        cfg.expected = std::string(argv[5]) == "mpc" ? &masp_mpc_parameters() : &e;
        try {
            auto p = LocalTxProver::from_paths(argv[2], argv[3], argv[4], cfg);
            std::printf("loaded\n");
        } catch (const Panic& x) {
            std::printf("panic at load: %s\n", x.what());
        }
        return 0;
    }
    if (argc != 3) {
        std::fprintf(stderr, "usage: %s <case.bin> <out.bin> | --selftest <file>\n", argv[0]);
        return 2;
    }
    Reader rd;
    {
        std::ifstream f(argv[1], std::ios::binary);
        rd.b.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    }
    if (rd.b.size() < 4 || std::memcmp(rd.b.data(), "MTP1", 4) != 0) {
        std::fprintf(stderr, "not a case file\n");
        return 2;
    }
    rd.at = 4;
    std::vector<uint8_t> params[3] = {rd.blob(), rd.blob(), rd.blob()};
    LocalTxProver::Config cfg;
    cfg.self_verify = rd.num<uint32_t>() != 0;
    cfg.threads = rd.num<uint32_t>();
    masp_hip_options opt;
    std::memset(&opt, 0, sizeof opt);
    opt.struct_size = sizeof opt;
    opt.batch_cap = (int32_t)rd.num<uint32_t>();
    cfg.options = &opt;
    cfg.expected = nullptr;  // the case's parameters are synthetic
    cfg.trace = std::getenv("MASP_TXP_TRACE") != nullptr;
    if (const char* v = std::getenv("MASP_TXP_DEVICES")) {  // "0,1,...": one prover over several GPUs (the same one twice on a one-GPU box)
        for (const char* p = v; *p;) {
            cfg.devices.push_back(std::atoi(p));
            while (*p && *p != ',') ++p;
            if (*p) ++p;
        }
        if (cfg.devices.size() > 1) opt.slots = 1, opt.bucket_tree_sub_batch = 32;  // (two contexts' scratch on one device in the test)
    }
    if (const char* v = std::getenv("MASP_TXP_CALLS")) cfg.calls_in_flight = (unsigned)std::atoi(v);
    const uint32_t mode = rd.num<uint32_t>(), n = rd.num<uint32_t>();
    std::vector<Record> recs(n);
    for (Record& r : recs) {
        r.kind = rd.num<uint32_t>();
        if (r.kind == MASP_HIP_SPEND) {
            SpendInfo& s = r.spend;
            s.proof_generation_key.ak = rd.bytes<32>();
            s.proof_generation_key.nsk = rd.bytes<32>();
            s.diversifier.bytes = rd.bytes<11>();
            s.rcm = rd.bytes<32>();
            s.ar = rd.bytes<32>();
            s.asset_type.identifier = rd.bytes<32>();
            s.value = rd.num<uint64_t>();
            s.anchor = rd.bytes<32>();
            s.merkle_path = rd.path();
            s.rcv = rd.bytes<32>();
        } else if (r.kind == MASP_HIP_OUTPUT) {
            OutputInfo& o = r.output;
            o.esk = rd.bytes<32>();
            o.payment_address.diversifier.bytes = rd.bytes<11>();
            o.payment_address.pk_d = rd.bytes<32>();
            o.rcm = rd.bytes<32>();
            o.asset_type.identifier = rd.bytes<32>();
            o.value = rd.num<uint64_t>();
            o.rcv = rd.bytes<32>();
        } else if (r.kind == MASP_HIP_CONVERT) {
            ConvertInfo& c = r.convert;
            c.allowed_conversion.generator = rd.bytes<32>();
            c.value = rd.num<uint64_t>();
            c.anchor = rd.bytes<32>();
            c.merkle_path = rd.path();
            c.rcv = rd.bytes<32>();
        } else {
            std::fprintf(stderr, "bad kind %u\n", r.kind);
            return 2;
        }
        r.rs.r = rd.bytes<32>();
        r.rs.s = rd.bytes<32>();
    }
    std::unique_ptr<LocalTxProver> prover;
    try {
        prover = LocalTxProver::from_bytes(params[0].data(), params[0].size(), params[1].data(), params[1].size(), params[2].data(), params[2].size(), cfg);
    } catch (const Panic& e) {
        std::printf("panic at load: %s\n", e.what());
        return 3;
    }
    std::printf("loaded: batch_cap %zu, host threads %u, devices %d\n", prover->batch_cap(), cfg.threads ? cfg.threads : detail::effective_cpus(),
                masp_hip_ctx_device_count(prover->context()));
    SaplingProvingContext ctx = prover->new_sapling_proving_context();
    size_t some = 0, none = 0, panics = 0;
    if (mode == 0) {
        for (Record& r : recs) {
            try {
                if (r.kind == MASP_HIP_SPEND) {
                    const SpendInfo& s = r.spend;
                    auto got = prover->spend_proof(ctx, s.proof_generation_key, s.diversifier, Rseed::before_zip212(s.rcm), s.ar, s.asset_type, s.value, s.anchor, s.merkle_path, s.rcv, &r.rs);
                    if (got) r.status = 1, r.zk = got->zkproof, r.cv = got->cv, r.rk = got->rk;
                } else if (r.kind == MASP_HIP_OUTPUT) {
                    const OutputInfo& o = r.output;
                    const ValueProof got = prover->output_proof(ctx, o.esk, o.payment_address, o.rcm, o.asset_type, o.value, o.rcv, &r.rs);
                    r.status = 1, r.zk = got.zkproof, r.cv = got.cv;
                } else {
                    const ConvertInfo& c = r.convert;
                    auto got = prover->convert_proof(ctx, c.allowed_conversion, c.value, c.anchor, c.merkle_path, c.rcv, &r.rs);
                    if (got) r.status = 1, r.zk = got->zkproof, r.cv = got->cv;
                }
            } catch (const Panic& e) {
                r.status = 2;
                std::printf("panic: %s\n", e.what());
            }
        }
    } else if (mode == 1) {
        std::vector<size_t> at[3];
        for (size_t i = 0; i < recs.size(); ++i) at[recs[i].kind].push_back(i);
        std::vector<SpendInfo> sp;
        std::vector<OutputInfo> ou;
        std::vector<ConvertInfo> co;
        std::vector<BlindingScalars> rs[3];
        for (int k = 0; k < 3; ++k)
            for (size_t i : at[k]) {
                rs[k].push_back(recs[i].rs);
                if (k == 0) sp.push_back(recs[i].spend);
                if (k == 1) ou.push_back(recs[i].output);
                if (k == 2) co.push_back(recs[i].convert);
            }
        try {
            prover->warm_up(sp.size(), ou.size(), co.size());  // every slot's scratch and the page-locked slabs, before the first real call
            size_t calls = 0, last = 0;
            const auto a = prover->spend_proofs(ctx, sp.data(), sp.size(), rs[0].data(), [&](size_t done, size_t total) {
                ++calls;
                last = done;
                (void)total;
            });
            std::printf("progress: %zu calls, last %zu of %zu\n", calls, last, sp.size());
            for (size_t q = 0; q < a.size(); ++q)
                if (a[q]) recs[at[0][q]].status = 1, recs[at[0][q]].zk = a[q]->zkproof, recs[at[0][q]].cv = a[q]->cv, recs[at[0][q]].rk = a[q]->rk;
            const auto b = prover->output_proofs(ctx, ou.data(), ou.size(), rs[1].data());
            for (size_t q = 0; q < b.size(); ++q)
                if (b[q]) recs[at[1][q]].status = 1, recs[at[1][q]].zk = b[q]->zkproof, recs[at[1][q]].cv = b[q]->cv;
            const auto c = prover->convert_proofs(ctx, co.data(), co.size(), rs[2].data());
            for (size_t q = 0; q < c.size(); ++q)
                if (c[q]) recs[at[2][q]].status = 1, recs[at[2][q]].zk = c[q]->zkproof, recs[at[2][q]].cv = c[q]->cv;
            // ... and the same prover shared by three threads, each with a context of its own (the reference's `&self` methods): four rounds of
            // the three calls per thread, in different orders — every result and every context must equal the ones above
            std::atomic<size_t> calls3{0}, diffs{0};
            auto same_sp = [](const std::optional<SpendProof>& x, const std::optional<SpendProof>& y) {
                return x.has_value() == y.has_value() && (!x || (x->zkproof == y->zkproof && x->cv == y->cv && x->rk == y->rk));
            };
            auto same_vp = [](const std::optional<ValueProof>& x, const std::optional<ValueProof>& y) {
                return x.has_value() == y.has_value() && (!x || (x->zkproof == y->zkproof && x->cv == y->cv));
            };
            auto user = [&](int t) {
                for (int round = 0; round < 4; ++round) {
                    SaplingProvingContext mine = prover->new_sapling_proving_context();
                    for (int step = 0; step < 3; ++step) {
                        const int k = (step + t + round) % 3;
                        if (k == 0) {
                            const auto x = prover->spend_proofs(mine, sp.data(), sp.size(), rs[0].data());
                            for (size_t q = 0; q < x.size(); ++q) diffs += !same_sp(x[q], a[q]);
                        } else if (k == 1) {
                            const auto x = prover->output_proofs(mine, ou.data(), ou.size(), rs[1].data());
                            for (size_t q = 0; q < x.size(); ++q) diffs += !same_vp(x[q], b[q]);
                        } else {
                            const auto x = prover->convert_proofs(mine, co.data(), co.size(), rs[2].data());
                            for (size_t q = 0; q < x.size(); ++q) diffs += !same_vp(x[q], c[q]);
                        }
                        ++calls3;
                    }
                    diffs += mine.bsk() != ctx.bsk() || mine.cv_sum() != ctx.cv_sum();
                }
            };
            std::vector<std::future<void>> users;
            for (int t = 0; t < 3; ++t) users.push_back(std::async(std::launch::async, user, t));
            for (auto& u : users) u.get();
            std::printf("shared by 3 threads: %zu calls, %zu differences\n", calls3.load(), diffs.load());
        } catch (const Panic& e) {
            std::printf("panic: %s\n", e.what());
            return 3;
        }
    }
    if (mode == 2) {
        // all Spend descriptions of the case through spend_proofs, blinding scalars from the system's generator, timed host to host
        // (synthesis and self-checks included): the first call of the prover after warm_up(), then a second one
        std::vector<SpendInfo> sp;
        for (const Record& r : recs)
            if (r.kind == MASP_HIP_SPEND) sp.push_back(r.spend);
        auto t0 = std::chrono::steady_clock::now();
        prover->warm_up(sp.size());
        std::printf("warm_up: %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        for (int pass = 0; pass < 2; ++pass) {
            SaplingProvingContext c2 = prover->new_sapling_proving_context();
            size_t calls = 0, last = 0;
            t0 = std::chrono::steady_clock::now();
            const auto got = prover->spend_proofs(c2, sp.data(), sp.size(), nullptr, [&](size_t done, size_t total) {
                ++calls;
                last = done == total ? done : last;
            });
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            size_t ok = 0;
            for (const auto& g : got) ok += g.has_value();
            std::printf("%s: %zu Spend descriptions, %zu proofs in %.3f s = %.1f proofs/s (progress called %zu times, last %zu)\n", pass ? "timed" : "first call",
                        sp.size(), ok, dt, ok / dt, calls, last);
        }
    }
    std::ofstream out(argv[2], std::ios::binary);
    for (const Record& r : recs) {
        some += r.status == 1;
        none += r.status == 0;
        panics += r.status == 2;
        const uint8_t st[4] = {(uint8_t)r.status, 0, 0, 0};
        out.write(reinterpret_cast<const char*>(st), 4);
        out.write(reinterpret_cast<const char*>(r.zk.data()), 192);
        out.write(reinterpret_cast<const char*>(r.cv.data()), 32);
        out.write(reinterpret_cast<const char*>(r.rk.data()), 32);
    }
    out.write(reinterpret_cast<const char*>(ctx.bsk().data()), 32);
    out.write(reinterpret_cast<const char*>(ctx.cv_sum().data()), 32);
    out.close();
    if (mode != 2) std::printf("descriptions %u: Some %zu, None %zu, Panic %zu\n", n, some, none, panics);  // (mode 2 keeps no results)
    return 0;
}
