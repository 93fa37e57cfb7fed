// bzk-worker: the proving worker of a Bazuka node as a NATIVE program over the C ABI of libbzk (include/bzk.h) - no Python, no torch.
//
// The loop the reference's external provers run:
//     POST /bincode/mpn/worker    PostMpnWorkerRequest{address}          -> PostMpnWorkerResponse{accepted}
//     GET  /bincode/mpn/work      GetMpnWorkRequest{address}             -> GetMpnWorkResponse{works: HashMap<usize, MpnWork>}
//     POST /bincode/mpn/solution  PostMpnSolutionRequest{prover, proofs} -> PostMpnSolutionResponse{accepted}
// (/root/reference/src/node/mod.rs:393-413, src/client/messages.rs:368-396, src/client/mod.rs:428-463; bodies are bincode 1.3, a GET
// carries its request in the body too).  A solution counts iff `MpnWork::verify` passes with the commitment bound to the prover's address
// and the work's reward (src/mpn/mod.rs:281-295).
//
// Everything that computes is libbzk: bzk_mpn_work_decode / _synthesize (host C++) and bzk_groth16_prove (HIP).  This file is the
// plumbing: an HTTP/1.1 client on POSIX sockets, the HashMap framing of the two messages, the proving-key sources, and the schedule -
// ONE producer thread synthesizes the witnesses of a round ahead of the proofs (bounded queue: a witness is 0.1 - 2 GB of pinned
// memory), one thread per prover SLOT takes them (slots of a device share its CRS: bzk_params_slot).  bazuka_amd/worker.py is the same
// loop for Python hosts; both are tested against tests/mock_node.py, which judges solutions with the oracle's pairing check.
//
// Proving keys: bellman `Parameters` files of the network (--params DEPOSIT WITHDRAW UPDATE; bzk_params_load_bellman) or the dev-mode
// CRS generated on the GPU from the circuit's matrices and a toxic-waste seed (--dev-toxic SEED; src/config/blockchain.rs:355-417;
// same derivation as worker.py: scalar i of kind k = ZkScalar::new(sha3_256("SEED/k/i") twice)).
// --dry-run: fetch, decode and synthesize only (no GPU context is created; nothing is posted) - checks a node's works against this
// build's circuits on any machine.
#include <arpa/inet.h>
#include <netdb.h>
#include <sys/socket.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "bzk.h"

namespace {

typedef std::vector<uint8_t> Bytes;
using clk = std::chrono::steady_clock;
double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

struct Fail : std::runtime_error {
    using std::runtime_error::runtime_error;
};
void ck(int32_t st, const char* what, bzk_ctx* ctx = nullptr) {
    if (st == BZK_OK) return;
    std::string m = std::string(what) + ": " + bzk_strerror(st);
    if (ctx) m += std::string(" (") + bzk_last_error(ctx) + ")";
    throw Fail(m);
}

// ---- HTTP/1.1 over a fresh connection per request (what http.client does for worker.py) ----------------------------------------------
struct Url { std::string host, port; };

int dial(const Url& u, double timeout_s) {
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_UNSPEC;
    hints.ai_socktype = SOCK_STREAM;
    if (getaddrinfo(u.host.c_str(), u.port.c_str(), &hints, &res) != 0 || !res) throw Fail("cannot resolve " + u.host);
    int fd = -1;
    for (addrinfo* a = res; a; a = a->ai_next) {
        fd = socket(a->ai_family, a->ai_socktype, a->ai_protocol);
        if (fd < 0) continue;
        timeval tv{(time_t)timeout_s, (suseconds_t)((timeout_s - (time_t)timeout_s) * 1e6)};
        setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
        if (connect(fd, a->ai_addr, a->ai_addrlen) == 0) break;
        close(fd);
        fd = -1;
    }
    freeaddrinfo(res);
    if (fd < 0) throw Fail("cannot connect to " + u.host + ":" + u.port);
    return fd;
}
void send_all(int fd, const void* p, size_t n) {
    const char* c = (const char*)p;
    while (n) {
        ssize_t k = send(fd, c, n, MSG_NOSIGNAL);
        if (k <= 0) throw Fail("send failed");
        c += k;
        n -= (size_t)k;
    }
}
// A response is bounded: a node (or whatever answers on its port) cannot make the worker buffer without limit.  The largest legitimate body is a
// round's works - a production Update work is ~0.6 MB of bincode - so 1 GiB is generous; the header block of any sane server fits 64 KiB.
constexpr size_t MAX_RESPONSE = (size_t)1 << 30, MAX_HEADER = (size_t)1 << 16;

Bytes http(const Url& u, const char* method, const char* path, const Bytes& body, double timeout_s) {
    const int fd = dial(u, timeout_s);
    struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
    char head[512];
    const int hl = snprintf(head, sizeof head,
                            "%s %s HTTP/1.1\r\nHost: %s:%s\r\nContent-Type: application/octet-stream\r\nContent-Length: %zu\r\nConnection: close\r\n\r\n",
                            method, path, u.host.c_str(), u.port.c_str(), body.size());
    send_all(fd, head, (size_t)hl);
    if (!body.empty()) send_all(fd, body.data(), body.size());
    Bytes in;
    char buf[1 << 16];
    size_t head_end = std::string::npos;
    long want = -1;  // Content-Length, -1 = until the peer closes
    bool chunked = false;
    int status = 0;
    for (;;) {
        if (head_end != std::string::npos && want >= 0 && in.size() - head_end >= (size_t)want) break;
        ssize_t k = recv(fd, buf, sizeof buf, 0);
        if (k < 0) throw Fail(std::string(method) + " " + path + ": receive failed (timeout?)");
        if (k == 0) break;
        if (in.size() + (size_t)k > MAX_RESPONSE) throw Fail(std::string(method) + " " + path + ": response larger than 1 GiB");
        const size_t scanned = in.size() < 3 ? 0 : in.size() - 3;  // a separator may straddle two reads
        in.insert(in.end(), buf, buf + k);
        if (head_end == std::string::npos) {
            static const char sep[4] = {'\r', '\n', '\r', '\n'};
            for (size_t i = scanned; i + 4 <= in.size(); ++i)
                if (!memcmp(&in[i], sep, 4)) { head_end = i + 4; break; }
            if (head_end == std::string::npos && in.size() > MAX_HEADER) throw Fail(std::string(method) + " " + path + ": no end of the HTTP header in 64 KiB");
            if (head_end != std::string::npos) {
                std::string h((const char*)in.data(), head_end);
                if (sscanf(h.c_str(), "HTTP/%*d.%*d %d", &status) != 1) throw Fail("malformed HTTP status line");
                for (char& ch : h) ch = (char)tolower((unsigned char)ch);
                const size_t p = h.find("\ncontent-length:");
                if (p != std::string::npos) {
                    want = atol(h.c_str() + p + 16);
                    if (want < 0 || (size_t)want > MAX_RESPONSE) throw Fail(std::string(method) + " " + path + ": implausible Content-Length");
                }
                chunked = h.find("transfer-encoding: chunked") != std::string::npos;
            }
        }
    }
    if (head_end == std::string::npos) throw Fail(std::string(method) + " " + path + ": no HTTP response");
    if (status != 200) throw Fail(std::string(method) + " " + path + ": HTTP " + std::to_string(status));
    Bytes payload(in.begin() + (long)head_end, in.end());
    if (want >= 0) {
        if (payload.size() < (size_t)want) throw Fail(std::string(method) + " " + path + ": short read");
        payload.resize((size_t)want);
    } else if (chunked) {
        Bytes out;
        size_t pos = 0;
        for (;;) {
            size_t e = pos;
            while (e + 1 < payload.size() && !(payload[e] == '\r' && payload[e + 1] == '\n')) ++e;
            if (e + 1 >= payload.size()) throw Fail("malformed chunked body");
            const unsigned long n = strtoul(std::string((const char*)&payload[pos], e - pos).c_str(), nullptr, 16);
            pos = e + 2;
            if (n == 0) break;
            if (n > payload.size() || pos + n + 2 > payload.size()) throw Fail("truncated chunked body");  // n first: pos + n must not wrap
            out.insert(out.end(), payload.begin() + (long)pos, payload.begin() + (long)(pos + n));
            pos += n + 2;
        }
        payload.swap(out);
    }
    return payload;
}

// ---- bincode framing of the three messages (the maps' payload types are decoded / encoded by libbzk) ----------------------------------
void put_u64(Bytes& b, uint64_t v) {
    for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(v >> (8 * i)));
}
uint64_t get_u64(const Bytes& b, size_t pos) {
    if (pos + 8 > b.size()) throw Fail("response truncated");
    uint64_t v = 0;
    for (int i = 0; i < 8; ++i) v |= (uint64_t)b[pos + i] << (8 * i);
    return v;
}
Bytes address_request(const uint8_t addr[32]) {  // GetMpnWorkRequest / PostMpnWorkerRequest: Address = a byte string of 32
    Bytes b;
    put_u64(b, 32);
    b.insert(b.end(), addr, addr + 32);
    return b;
}

struct WorkDel { void operator()(bzk_mpn_work* w) const { bzk_mpn_work_free(w); } };
struct R1csDel { void operator()(bzk_r1cs* r) const { bzk_r1cs_free(r); } };
typedef std::unique_ptr<bzk_mpn_work, WorkDel> WorkPtr;
typedef std::unique_ptr<bzk_r1cs, R1csDel> R1csPtr;

struct Work {
    uint64_t id = 0;
    WorkPtr w;
    uint64_t info[12] = {};
    int kind() const { return (int)info[0]; }
};

std::vector<Work> parse_works(const Bytes& body, uint32_t flags) {
    std::vector<Work> out;
    const uint64_t n = get_u64(body, 0);
    if (n > body.size()) throw Fail("work response: implausible count");
    size_t pos = 8;
    for (uint64_t i = 0; i < n; ++i) {
        Work x;
        x.id = get_u64(body, pos);
        pos += 8;
        bzk_mpn_work* w = nullptr;
        uint64_t used = 0;
        const int32_t st = bzk_mpn_work_decode(body.data() + pos, body.size() - pos, flags, &w, &used);
        if (st != BZK_OK) throw Fail(std::string("work response: ") + bzk_mpn_work_last_error());
        x.w.reset(w);
        ck(bzk_mpn_work_info(w, x.info), "bzk_mpn_work_info");
        pos += used;
        out.push_back(std::move(x));
    }
    if (pos != body.size()) throw Fail("work response: trailing bytes");
    return out;
}

// ---- proving keys -----------------------------------------------------------------------------------------------------------------------
R1csPtr shape_circuit(int kind, uint32_t L, uint32_t T, uint32_t B, bool matrices) {
    static const uint8_t z[32] = {0};
    bzk_r1cs* r = nullptr;
    if (kind == 2)
        ck(bzk_mpn_update_empty(L, T, B, z, 0, z, z, z, z, matrices ? 1 : 0, &r), "bzk_mpn_update_empty");
    else
        ck(bzk_mpn_circuit_empty(kind, L, T, B, z, 0, z, z, z, matrices ? 1 : 0, &r), "bzk_mpn_circuit_empty");
    return R1csPtr(r);
}
template <class T>
const T* view(const bzk_r1cs* r, int which, uint64_t* bytes = nullptr) {
    uint64_t n = 0;
    const void* p = bzk_r1cs_data(r, which, &n);
    if (bytes) *bytes = n;
    return (const T*)p;
}

struct Key {
    bzk_params* params = nullptr;
    Bytes vk;  // bincode(Groth16VerifyingKey)
};
struct KeySource {
    bzk_ctx* ctx;
    std::string dev_toxic;     // dev mode when non-empty
    std::string files[3];      // bellman Parameters files otherwise
    std::map<std::vector<uint32_t>, Key> cache;
    std::mutex m;

    void toxic(int kind, uint8_t out[160]) const {
        for (int i = 0; i < 5; ++i) {
            const std::string s = dev_toxic + "/" + std::to_string(kind) + "/" + std::to_string(i);
            uint8_t d[64];
            ck(bzk_host_sha3_256((const uint8_t*)s.data(), s.size(), d), "bzk_host_sha3_256");
            memcpy(d + 32, d, 32);
            ck(bzk_host_scalar_new(d, 64, out + 32 * i), "bzk_host_scalar_new");
        }
    }
    const Key& get(const Work& w) {
        const std::vector<uint32_t> shape{(uint32_t)w.info[0], (uint32_t)w.info[1], (uint32_t)w.info[2], (uint32_t)w.info[3]};
        std::lock_guard<std::mutex> g(m);
        auto it = cache.find(shape);
        if (it != cache.end()) return it->second;
        Key k;
        k.vk.resize(878 + 97 * 16);
        const auto t0 = clk::now();
        if (!dev_toxic.empty()) {
            R1csPtr r = shape_circuit((int)shape[0], shape[1], shape[2], shape[3], true);
            uint64_t info[9];
            ck(bzk_r1cs_info(r.get(), info), "bzk_r1cs_info");
            bzk_csr m3[3];
            for (int x = 0; x < 3; ++x)
                m3[x] = {info[2], view<uint32_t>(r.get(), 12 + x), view<uint32_t>(r.get(), 9 + x), view<uint8_t>(r.get(), 6 + x)};
            uint8_t tox[160];
            toxic((int)shape[0], tox);
            ck(bzk_groth16_setup(ctx, &m3[0], &m3[1], &m3[2], (uint32_t)info[0], (uint32_t)info[1], tox, &k.params, k.vk.data(), k.vk.size()),
               "bzk_groth16_setup", ctx);
            k.vk.resize(878 + 97 * info[0]);
        } else {
            // the density maps (views 4, 5) come with the matrices: the file itself does not say which variables a / b cover
            R1csPtr r = shape_circuit((int)shape[0], shape[1], shape[2], shape[3], true);
            uint64_t info[9];
            ck(bzk_r1cs_info(r.get(), info), "bzk_r1cs_info");
            std::ifstream f(files[shape[0]], std::ios::binary);
            if (!f) throw Fail("cannot open " + files[shape[0]]);
            Bytes blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            ck(bzk_params_load_bellman(ctx, blob.data(), blob.size(), (uint32_t)info[0], (uint32_t)info[1], view<uint8_t>(r.get(), 4),
                                       view<uint8_t>(r.get(), 5), &k.params, k.vk.data(), k.vk.size()),
               "bzk_params_load_bellman", ctx);
            k.vk.resize(878 + 97 * info[0]);
        }
        fprintf(stderr, "[bzk-worker] proving key for kind %u (L=%u, T=%u, B=%u) ready in %.1f s\n", shape[0], shape[1], shape[2], shape[3],
                secs(t0, clk::now()));
        return cache.emplace(shape, std::move(k)).first->second;
    }
    void close() {
        for (auto& kv : cache) bzk_params_free(ctx, kv.second.params);
        cache.clear();
    }
};

// a prover slot: its own context; slot 0 of a device owns the keys, the others share them through bzk_params_slot
struct Slot {
    bzk_ctx* ctx = nullptr;
    KeySource* keys = nullptr;                       // the device's key source
    bool first = false;
    std::map<const bzk_params*, bzk_params*> mine;   // further slots: the first slot's handle -> this slot's
    bzk_params* params_for(const Work& w, const Bytes& work_vk) {
        const Key& k = keys->get(w);
        if (k.vk != work_vk) throw Fail("the work's verifying key is not the one of this worker's proving key");
        if (first) return k.params;
        auto it = mine.find(k.params);
        if (it != mine.end()) return it->second;
        bzk_params* p = nullptr;
        ck(bzk_params_slot(ctx, k.params, &p), "bzk_params_slot", ctx);
        mine[k.params] = p;
        return p;
    }
};

struct Stats {
    std::mutex m;
    uint64_t fetched = 0, proved = 0, accepted = 0, unsat = 0, self_check_failed = 0, errors = 0, rounds = 0;
    double synth_s = 0, prove_s = 0;
    std::string last_error;
    std::vector<uint64_t> by_slot;
};

struct Options {
    Url node;
    uint8_t address[32];
    std::string dev_toxic, files[3];
    std::vector<int> devices{0};
    int slots_per_device = 1, threads = 0;
    double poll = 1.0, timeout_s = 30.0;
    long rounds = -1;
    bool self_check = false, dry_run = false;
    bool defer = false;  // --defer: BZK_SYNTH_DEFER + bzk_groth16_prove_r1cs (the hash-dependent witness values on the device: less host CPU per work)
    uint32_t flags = 0;
};

struct Item {
    Work* work;
    R1csPtr r1cs;
    double synth_s;
};

void random_scalar(uint8_t out[32]) {  // bellman: `E::Fr::random(rng)`
    uint8_t raw[64];
    std::ifstream f("/dev/urandom", std::ios::binary);
    f.read((char*)raw, 64);
    if (!f) throw Fail("cannot read /dev/urandom");
    ck(bzk_host_scalar_new(raw, 64, out), "bzk_host_scalar_new");
}

// one round: fetch -> (synthesize ahead | prove on the slots) -> post.  Returns `accepted`.
uint64_t run_once(const Options& o, std::vector<Slot>& slots, Stats& st) {
    const Bytes req = address_request(o.address);
    std::vector<Work> works = parse_works(http(o.node, "GET", "/bincode/mpn/work", req, o.timeout_s), o.flags);
    {
        std::lock_guard<std::mutex> g(st.m);
        st.fetched += works.size();
    }
    if (works.empty()) return 0;
    const size_t n_slots = o.dry_run ? 1 : std::min(slots.size(), works.size());
    std::map<uint64_t, Bytes> work_vk;
    for (auto& w : works) {
        Bytes vk(878 + 97 * 16);
        uint64_t len = 0;
        ck(bzk_mpn_work_vk(w.w.get(), -1, vk.data(), vk.size(), &len), "bzk_mpn_work_vk");
        vk.resize(len);
        work_vk[w.id] = std::move(vk);
    }
    // keys are generated / loaded on the first slot's context of each device BEFORE the slot threads start: no context is ever used
    // from two threads
    if (!o.dry_run)
        for (auto& w : works)
            for (auto& s : slots)
                if (s.first) (void)s.keys->get(w);

    std::mutex qm;
    std::condition_variable qcv;
    std::deque<Item> ready;
    bool produced_all = false;
    std::map<uint64_t, Bytes> proofs;
    auto failed = [&](const std::string& where, uint64_t wid, const std::string& what) {
        std::lock_guard<std::mutex> g(st.m);
        ++st.errors;
        st.last_error = where + ", work " + std::to_string(wid) + ": " + what;
        fprintf(stderr, "[bzk-worker] %s\n", st.last_error.c_str());
    };
    std::thread producer([&] {
        for (auto& w : works) {
            const auto t0 = clk::now();
            bzk_r1cs* r = nullptr;
            const int32_t s = bzk_mpn_work_synthesize(w.w.get(), o.address, nullptr, o.threads, o.defer ? BZK_SYNTH_DEFER : 0, &r);
            if (s != BZK_OK) {
                failed("synthesis", w.id, bzk_strerror(s));
                continue;
            }
            std::unique_lock<std::mutex> lk(qm);
            qcv.wait(lk, [&] { return ready.size() < n_slots; });  // at most one witness per slot waits
            ready.push_back({&w, R1csPtr(r), secs(t0, clk::now())});
            qcv.notify_all();
        }
        std::lock_guard<std::mutex> lk(qm);
        produced_all = true;
        qcv.notify_all();
    });
    auto consumer = [&](size_t si) {
        for (;;) {
            Item it;
            {
                std::unique_lock<std::mutex> lk(qm);
                qcv.wait(lk, [&] { return !ready.empty() || produced_all; });
                if (ready.empty()) return;
                it = std::move(ready.front());
                ready.pop_front();
                qcv.notify_all();
            }
            try {
                uint64_t info[9];
                ck(bzk_r1cs_info(it.r1cs.get(), info), "bzk_r1cs_info");
                if (info[6] != 0) {  // the witness does not satisfy its circuit: a proof of it could only be refused by the node
                    std::lock_guard<std::mutex> g(st.m);
                    st.synth_s += it.synth_s;
                    ++st.unsat;
                    continue;
                }
                if (o.dry_run) {
                    std::lock_guard<std::mutex> g(st.m);
                    st.synth_s += it.synth_s;
                    continue;
                }
                Slot& s = slots[si];
                const auto t1 = clk::now();
                bzk_params* ph = s.params_for(*it.work, work_vk.at(it.work->id));  // at(): a lookup only - the map is shared by the slot threads
                bzk_assignment a{};
                uint64_t zb = 0;
                a.z = view<uint8_t>(it.r1cs.get(), 0, &zb);
                a.az = view<uint8_t>(it.r1cs.get(), 1);
                a.bz = view<uint8_t>(it.r1cs.get(), 2);
                a.cz = view<uint8_t>(it.r1cs.get(), 3);
                a.n_rows = info[2];
                a.n_vars = zb / 32;
                uint8_t r[32], sb[32];
                random_scalar(r);
                random_scalar(sb);
                Bytes proof(387);
                if (o.defer) {  // completes a deferred instance on the device first; a complete one goes through unchanged
                    const int32_t pst = bzk_groth16_prove_r1cs(s.ctx, ph, it.r1cs.get(), r, sb, proof.data());
                    if (pst == BZK_E_UNSAT) {  // a violated DEFERRED row: only the device-side fill sees it - the same outcome as info[6] != 0 above
                        std::lock_guard<std::mutex> g(st.m);
                        st.synth_s += it.synth_s;
                        st.prove_s += secs(t1, clk::now());
                        ++st.unsat;
                        continue;
                    }
                    ck(pst, "bzk_groth16_prove_r1cs", s.ctx);
                } else
                    ck(bzk_groth16_prove(s.ctx, ph, &a, r, sb, proof.data()), "bzk_groth16_prove", s.ctx);
                const bool ok = !o.self_check || bzk_mpn_work_verify(it.work->w.get(), o.address, proof.data()) == 1;
                std::lock_guard<std::mutex> g(st.m);
                st.synth_s += it.synth_s;
                st.prove_s += secs(t1, clk::now());
                ++st.proved;
                ++st.by_slot[si];
                if (ok)
                    proofs[it.work->id] = std::move(proof);
                else
                    ++st.self_check_failed;
            } catch (const std::exception& e) {
                failed("slot " + std::to_string(si), it.work->id, e.what());
            }
        }
    };
    std::vector<std::thread> th;
    for (size_t i = 1; i < n_slots; ++i) th.emplace_back(consumer, i);
    consumer(0);
    for (auto& t : th) t.join();
    producer.join();
    if (o.dry_run || proofs.empty()) return 0;
    Bytes body;  // PostMpnSolutionRequest{prover, proofs: HashMap<usize, ZkProof>}
    put_u64(body, 32);
    body.insert(body.end(), o.address, o.address + 32);
    put_u64(body, proofs.size());
    for (auto& kv : proofs) {
        put_u64(body, kv.first);
        uint8_t enc[391];
        ck(bzk_zkproof_encode(kv.second.data(), enc), "bzk_zkproof_encode");
        body.insert(body.end(), enc, enc + 391);
    }
    const Bytes resp = http(o.node, "POST", "/bincode/mpn/solution", body, o.timeout_s);
    if (resp.size() != 8) throw Fail("solution response: expected 8 bytes");
    const uint64_t acc = get_u64(resp, 0);
    std::lock_guard<std::mutex> g(st.m);
    st.accepted += acc;
    return acc;
}

double cpu_quota() {  // CPUs this container may use (cgroup v2 cpu.max / v1 cfs quota); <= 0 when unlimited
    std::ifstream f("/sys/fs/cgroup/cpu.max");
    std::string q;
    double per = 0;
    if (f >> q >> per) return (q == "max" || per <= 0) ? 0 : atof(q.c_str()) / per;
    std::ifstream a("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), b("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
    double qq = 0;
    if ((a >> qq) && (b >> per) && qq > 0 && per > 0) return qq / per;
    return 0;
}

bool hex32(const char* s, uint8_t out[32]) {
    if (strlen(s) != 64) return false;
    for (int i = 0; i < 32; ++i) {
        unsigned v;
        if (sscanf(s + 2 * i, "%2x", &v) != 1) return false;
        out[i] = (uint8_t)v;
    }
    return true;
}

int usage(const char* why) {
    fprintf(stderr,
            "%s\nusage: bzk-worker --node HOST:PORT --address <64 hex> (--dev-toxic SEED | --params DEPOSIT WITHDRAW UPDATE)\n"
            "                  [--devices 0,1,..] [--slots-per-device N] [--threads N] [--poll S] [--rounds N] [--timeout S]\n"
            "                  [--self-check] [--defer] [--sig-len-prefixed] [--dry-run]\n",
            why);
    return 2;
}

}  // namespace

int main(int argc, char** argv) {
    // 16 hardware queues for the slots' streams instead of HIP's default 4 (read by the runtime at its first call; a value set by the operator wins):
    // bazuka_amd/__init__.py has the measurement
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    // launches through the runtime's per-stream worker threads instead of from the calling threads (round 6, runs 37 - 39): the prover side's host CPU per proof
    // 0.0174 -> 0.0065 CPU-s (the runtime's helper threads no longer spend 0.009 s of system time per proof), four slots + 4 % proofs/s under a CPU quota
    // (profiles/r06_run37_39_host_cpu_of_the_prover.txt); read by the runtime when it initialises, an operator's own setting wins
    setenv("AMD_DIRECT_DISPATCH", "0", 0);
    Options o;
    bool have_node = false, have_addr = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--node") {
            const std::string v = next();
            const size_t c = v.rfind(':');
            if (c == std::string::npos) return usage("--node wants HOST:PORT");
            o.node = {v.substr(0, c), v.substr(c + 1)};
            have_node = true;
        } else if (a == "--address") {
            if (!(have_addr = hex32(next(), o.address))) return usage("--address must be 32 bytes of hex");
        } else if (a == "--dev-toxic") {
            o.dev_toxic = next();
        } else if (a == "--params") {
            for (int k = 0; k < 3; ++k) o.files[k] = next();
        } else if (a == "--devices" || a == "--device") {
            o.devices.clear();
            std::string v = next();
            for (size_t p = 0; p <= v.size();) {
                const size_t q = v.find(',', p);
                o.devices.push_back(atoi(v.substr(p, q == std::string::npos ? q : q - p).c_str()));
                if (q == std::string::npos) break;
                p = q + 1;
            }
        } else if (a == "--slots-per-device") {
            o.slots_per_device = std::max(1, atoi(next()));
        } else if (a == "--threads") {
            o.threads = atoi(next());
        } else if (a == "--poll") {
            o.poll = atof(next());
        } else if (a == "--rounds") {
            o.rounds = atol(next());
        } else if (a == "--timeout") {
            o.timeout_s = atof(next());
        } else if (a == "--defer") {
            o.defer = true;
        } else if (a == "--self-check") {
            o.self_check = true;
        } else if (a == "--sig-len-prefixed") {
            o.flags |= BZK_WORK_SIG_LEN_PREFIXED;
        } else if (a == "--dry-run") {
            o.dry_run = true;
        } else {
            return usage(("unknown option " + a).c_str());
        }
    }
    if (!have_node || !have_addr) return usage("--node and --address are required");
    if (!o.dry_run && (o.dev_toxic.empty() == o.files[2].empty())) return usage("exactly one of --dev-toxic / --params");
    // A prover slot keeps ~5 host threads waiting on the GPU.  Under a CPU quota smaller than the threads this worker runs, spinning waits
    // are charged against the budget the witness generation needs: the waits sleep on interrupts instead (libbzk reads BZK_SYNC_BLOCKING
    // once, when the first context is created).
    const double quota = cpu_quota();
    if (quota > 0 && quota < (double)(o.devices.size() * (size_t)o.slots_per_device * 5 + std::thread::hardware_concurrency() / 2))
        setenv("BZK_SYNC_BLOCKING", "1", 0);

    std::vector<Slot> slots;
    std::vector<std::unique_ptr<KeySource>> sources;
    Stats st;
    int rc = 0;
    try {
        if (!o.dry_run) {
            for (int d : o.devices) {
                for (int k = 0; k < o.slots_per_device; ++k) {
                    Slot s;
                    ck(bzk_ctx_create(d, nullptr, &s.ctx), "bzk_ctx_create (no usable gfx950 device? there is no CPU prover)");
                    if (k == 0) {
                        sources.emplace_back(new KeySource);
                        sources.back()->ctx = s.ctx;
                        sources.back()->dev_toxic = o.dev_toxic;
                        for (int x = 0; x < 3; ++x) sources.back()->files[x] = o.files[x];
                        s.first = true;
                    }
                    s.keys = sources.back().get();
                    slots.push_back(std::move(s));
                }
            }
        }
        st.by_slot.assign(std::max<size_t>(1, slots.size()), 0);
        const Bytes reg = http(o.node, "POST", "/bincode/mpn/worker", address_request(o.address), o.timeout_s);
        fprintf(stderr, "[bzk-worker] registered: %s\n", (reg.size() == 1 && reg[0] == 1) ? "accepted" : "refused");
        for (long done = 0; o.rounds < 0 || done < o.rounds; ++done) {
            try {
                (void)run_once(o, slots, st);
            } catch (const std::exception& e) {  // node away, a short read or a bad payload: keep polling, as a worker daemon does
                std::lock_guard<std::mutex> g(st.m);
                st.last_error = e.what();
                ++st.errors;
                fprintf(stderr, "[bzk-worker] round failed: %s\n", e.what());
            }
            ++st.rounds;
            if (o.rounds < 0 || done + 1 < o.rounds) std::this_thread::sleep_for(std::chrono::duration<double>(o.poll));
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "[bzk-worker] fatal: %s\n", e.what());
        st.last_error = e.what();
        rc = 1;
    }
    std::string by = "[";
    for (size_t i = 0; i < st.by_slot.size(); ++i) by += (i ? ", " : "") + std::to_string(st.by_slot[i]);
    by += "]";
    std::string err = st.last_error;
    for (char& c : err)
        if (c == '"' || c == '\\' || c == '\n') c = ' ';
    printf("{\"rounds\": %llu, \"fetched\": %llu, \"proved\": %llu, \"accepted\": %llu, \"unsat\": %llu, \"self_check_failed\": %llu, \"errors\": %llu, "
           "\"synth_s\": %.3f, \"prove_s\": %.3f, \"proved_by_slot\": %s, \"dry_run\": %s, \"last_error\": \"%s\"}\n",
           (unsigned long long)st.rounds, (unsigned long long)st.fetched, (unsigned long long)st.proved, (unsigned long long)st.accepted,
           (unsigned long long)st.unsat, (unsigned long long)st.self_check_failed, (unsigned long long)st.errors, st.synth_s, st.prove_s, by.c_str(),
           o.dry_run ? "true" : "false", err.c_str());
    // slots go before the keys they share
    for (auto& s : slots)
        for (auto& kv : s.mine) bzk_params_free(s.ctx, kv.second);
    for (auto& src : sources) src->close();
    for (auto& s : slots) bzk_ctx_destroy(s.ctx);
    return rc;
}
