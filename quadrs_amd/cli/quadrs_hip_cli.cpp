// quadrs-hip — the reference's CLI operator chain (from / gen / shift / lowpass / sparkfft / bucket /
// write) driven through the C ABI of the MI355X engine (include/quadrs_hip.h).
//
// Host glue only: the grammar and defaults restate src/args.rs, the chain plumbing restates
// Operation::exec (src/lib.rs:83-175) and the Samples iterator (src/samples.rs:11-28).  Every
// sample-touching step is a call into libquadrs_hip.so; there is no CPU arithmetic path here.
// stdout is byte-identical to the reference's for the same command line; diagnostics go to stderr.
//
// Chains of the shape  from|gen [shift] [lowpass]  ->  sparkfft|bucket  run as ONE fused plan
// (qd_plan_*).  Anything else (e.g. two lowpasses, write) falls back to the block iterator, whose
// read_at() calls the fine-grained entry points exactly where the reference's read_at() computes.
#include <cerrno>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <map>
#include <memory>
#include <regex>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

#include "../../include/quadrs_hip.h"

namespace {

struct Fail { std::string msg; };
[[noreturn]] void bail(const std::string &m) { throw Fail{m}; }
void qd_check(int rc, const char *what) {
    if (rc != QD_OK) bail(std::string(what) + ": " + qd_last_error() + (rc == QD_ERR_PANIC ? " (the reference panics here)" : ""));
}

// ------------------------------------------------------------------ src/args.rs

// find_multiplication_suffix + parse_si_*  (src/args.rs:335-379)
bool split_si(const std::string &from, std::string &val, uint64_t &mul) {
    mul = 1; val = from;
    if (from.empty()) return true;
    switch (from.back()) {
    case 'k': mul = 1000ull; break;
    case 'M': mul = 1000000ull; break;
    case 'G': mul = 1000000000ull; break;
    default: return true;
    }
    val = from.substr(0, from.size() - 1);
    return true;
}
uint64_t parse_si_u64(const std::string &s) {
    std::string v; uint64_t mul; split_si(s, v, mul);
    if (v.empty() || v.find_first_not_of("0123456789") != std::string::npos) bail("invalid digit found in string: " + s);
    errno = 0;
    unsigned long long p = strtoull(v.c_str(), nullptr, 10);
    if (errno) bail("number too large: " + s);
    unsigned __int128 r = (unsigned __int128)p * mul;
    if (r > UINT64_MAX) bail("unit is out of range: " + s);
    return (uint64_t)r;
}
int64_t parse_si_i64(const std::string &s) {
    std::string v; uint64_t mul; split_si(s, v, mul);
    size_t start = (!v.empty() && (v[0] == '-' || v[0] == '+')) ? 1 : 0;
    if (v.size() == start || v.find_first_not_of("0123456789", start) != std::string::npos) bail("invalid digit found in string: " + s);
    errno = 0;
    long long p = strtoll(v.c_str(), nullptr, 10);
    if (errno) bail("number too large: " + s);
    __int128 r = (__int128)p * (__int128)mul;
    if (r > INT64_MAX || r < INT64_MIN) bail("unit is out of range: " + s);
    return (int64_t)r;
}
double parse_si_f64(const std::string &s) {
    std::string v; uint64_t mul; split_si(s, v, mul);
    char *end = nullptr;
    double p = strtod(v.c_str(), &end);
    if (v.empty() || *end) bail("invalid float literal: " + s);
    return p * (double)mul;
}
bool parse_bool(const std::string &s) {       // src/args.rs:381-390
    if (s == "true" || s == "yes" || s == "y") return true;
    if (s == "false" || s == "no" || s == "n") return false;
    bail("unacceptable boolean value: '" + s + "'");
}

// guess_from_extension (src/args.rs:392-402)
int format_from_ext(const std::string &ext) {
    if (ext == "cf32" || ext == "fc32") return QD_FMT_CF32;
    if (ext == "cs8" || ext == "sc8" || ext == "c8") return QD_FMT_CS8;
    if (ext == "cu8" || ext == "su8") return QD_FMT_CU8;
    if (ext == "cs16" || ext == "sc16" || ext == "c16") return QD_FMT_CS16;
    return -1;
}

enum OpKind { OP_FROM, OP_GEN, OP_SHIFT, OP_LOWPASS, OP_SPARKFFT, OP_BUCKET, OP_WRITE };
struct Op {
    OpKind kind;
    std::string filename; int format = 0; uint64_t sample_rate = 0;     // from
    std::vector<int64_t> cos; double seconds = 1.0;                     // gen
    int64_t shift = 0;                                                  // shift
    uint64_t lp_freq = 0, decimate = 8; size_t size = 40;               // lowpass
    size_t width = 128; uint64_t stride = 128; bool has_range = false; float rmin = 0, rmax = 0;   // sparkfft / bucket
    size_t levels = 2;
    bool overwrite = false; std::string prefix;                         // write
};

typedef std::map<std::string, std::vector<std::string>> ArgMap;

// read_just_args (src/args.rs:404-445): "-name value" pairs; a token whose third char is a digit is a number
ArgMap read_just_args(const std::vector<std::string> &argv, size_t &i) {
    ArgMap ret;
    while (i < argv.size()) {
        const std::string &opt = argv[i];
        if (opt.empty() || opt[0] != '-') break;
        if (opt.size() > 2 && isdigit((unsigned char)opt[2])) break;
        ++i;
        if (i >= argv.size()) bail(opt + " requires an argument");
        if (argv[i].empty()) bail(opt + " requires a non-empty argument");
        ret[opt.substr(1)].push_back(argv[i]);
        ++i;
    }
    return ret;
}
std::map<std::string, std::string> no_duplicates(const ArgMap &m) {     // src/args.rs:447-454
    std::map<std::string, std::string> r;
    for (auto &kv : m) {
        if (kv.second.size() != 1) bail("'-" + kv.first + "' specified more than once");
        r[kv.first] = kv.second[0];
    }
    return r;
}
std::string take(std::map<std::string, std::string> &m, const char *k, bool *found) {
    auto it = m.find(k);
    if (it == m.end()) { *found = false; return ""; }
    std::string v = it->second; m.erase(it); *found = true; return v;
}
void ensure_empty(const std::map<std::string, std::string> &m) {
    if (!m.empty()) bail("invalid flags: [\"" + m.begin()->first + "\"]");
}

// guess_details / guess_format_from_name (src/args.rs:65-135,328-333)
void guess_details(const std::string &filename, const std::string *sr_override, const std::string *fmt_override,
                   uint64_t &sample_rate, int &format) {
    std::string sr;
    int fmt = -1;
    std::smatch m;
    if (std::regex_search(filename, m, std::regex("\\bsr([0-9]+[kMG]?)\\b"))) sr = m[1];
    if (std::regex_search(filename, m, std::regex("gqrx_.*?_[0-9]+_([0-9]+)_fc.raw"))) { sr = m[1]; fmt = QD_FMT_CF32; }
    if (std::regex_search(filename, m, std::regex("g\\d+_\\d+(?:\\.\\d+)?M_(\\d+k).cu8"))) { sr = m[1]; fmt = QD_FMT_CU8; }
    size_t dot = filename.rfind('.');
    if (dot != std::string::npos) { int g = format_from_ext(filename.substr(dot + 1)); if (g >= 0) fmt = g; }
    if (sr_override) sr = *sr_override;
    if (fmt_override) { fmt = format_from_ext(*fmt_override); if (fmt < 0) bail("unrecognised extension: \"" + *fmt_override + "\""); }
    if (sr.empty()) bail("unable to guess sample rate from filename \"" + filename + "\", please specify it");
    if (fmt < 0) bail("unable to guess format from filename \"" + filename + "\", please specify it");
    sample_rate = parse_si_u64(sr);
    format = fmt;
}

std::vector<Op> parse(const std::vector<std::string> &argv) {          // src/args.rs:19-45
    std::vector<Op> ops;
    size_t i = 0;
    while (i < argv.size()) {
        std::string cmd = argv[i++];
        ArgMap raw = read_just_args(argv, i);
        Op op{};
        auto next = [&](const char *err) -> std::string { if (i >= argv.size()) bail(err); return argv[i++]; };
        bool f;
        if (cmd == "from") {
            auto m = no_duplicates(raw);
            std::string fn = next("'from' requires a filename argument");
            std::string sr = take(m, "sr", &f); bool has_sr = f;
            std::string fm = take(m, "format", &f); bool has_fm = f;
            ensure_empty(m);
            op.kind = OP_FROM; op.filename = fn;
            guess_details(fn, has_sr ? &sr : nullptr, has_fm ? &fm : nullptr, op.sample_rate, op.format);
        } else if (cmd == "shift") {
            auto m = no_duplicates(raw);
            if (!m.empty()) bail("'shift' has no named arguments");
            op.kind = OP_SHIFT; op.shift = parse_si_i64(next("'shift' requires a frequency argument"));
        } else if (cmd == "lowpass") {
            auto m = no_duplicates(raw);
            op.kind = OP_LOWPASS;
            op.lp_freq = parse_si_u64(next("'lowpass' requires a frequency argument"));
            std::string v = take(m, "power", &f);
            op.size = f ? (size_t)parse_si_u64(v) * 2 : 40;              // src/args.rs:161-166
            v = take(m, "decimate", &f);
            op.decimate = f ? parse_si_u64(v) : 8;                       // :168-171
            ensure_empty(m);
        } else if (cmd == "sparkfft") {
            auto m = no_duplicates(raw);
            op.kind = OP_SPARKFFT;
            std::string v = take(m, "width", &f);
            op.width = f ? (size_t)parse_si_u64(v) : 128;                // :186-189
            v = take(m, "stride", &f);
            op.stride = f ? parse_si_u64(v) : op.width;                  // :191-194
            v = take(m, "range", &f);
            if (f) {                                                     // :196-207
                size_t c = v.find(':');
                if (c == std::string::npos) bail("range argument must contain a ':': '" + v + "'");
                op.has_range = true;
                op.rmin = strtof(v.substr(0, c).c_str(), nullptr);
                op.rmax = strtof(v.substr(c + 1).c_str(), nullptr);
            }
            ensure_empty(m);
        } else if (cmd == "bucket") {
            auto m = no_duplicates(raw);
            op.kind = OP_BUCKET;
            std::string lv = next("bucket usage: bucket -by freq [number-of-buckets]");
            op.levels = (size_t)strtoull(lv.c_str(), nullptr, 10);
            std::string v = take(m, "width", &f);
            op.width = f ? (size_t)parse_si_u64(v) : 128;
            v = take(m, "stride", &f);
            op.stride = f ? parse_si_u64(v) : op.width;
            v = take(m, "by", &f);
            if (!f || v != "freq") bail("must bucket -by freq");
            ensure_empty(m);
        } else if (cmd == "write") {
            auto m = no_duplicates(raw);
            op.kind = OP_WRITE;
            std::string v = take(m, "overwrite", &f);
            op.overwrite = f ? parse_bool(v) : false;
            ensure_empty(m);
            op.prefix = next("'write' requires a prefix argument");
        } else if (cmd == "gen") {
            op.kind = OP_GEN;
            auto it = raw.find("cos");
            if (it == raw.end()) bail("gen requires at least one operation");
            for (auto &s : it->second) op.cos.push_back(parse_si_i64(s));
            raw.erase(it);
            auto il = raw.find("len");
            if (il != raw.end()) {
                if (il->second.size() != 1) bail("len requires exactly one value");
                op.seconds = parse_si_f64(il->second[0]);
                raw.erase(il);
            }
            if (!raw.empty()) bail("invalid flags: [\"" + raw.begin()->first + "\"]");
            op.sample_rate = parse_si_u64(next("sample rate argument required"));
        } else if (cmd == "ui" || cmd == "eui") {
            bail("'" + cmd + "' (GUI) is out of scope of the MI355X engine");
        } else {
            bail("processing command: \"" + cmd + "\": unrecognised command");
        }
        ops.push_back(op);
    }
    return ops;
}

// ------------------------------------------------------------------ Samples iterator (src/samples.rs:11-28)

const size_t PANIC = (size_t)-1;

struct Samples {
    virtual ~Samples() {}
    virtual uint64_t len() const = 0;
    virtual uint64_t sample_rate() const = 0;
    virtual size_t read_at(uint64_t off, qd_c32 *buf, size_t n) const = 0;
    void read_exact_at(uint64_t off, qd_c32 *buf, size_t n) const {
        size_t got = read_at(off, buf, n);
        if (got != n) bail("TODO: read-exact messed up: " + std::to_string(n) + " (wanted) != " + std::to_string(got) +
                           " (read) at " + std::to_string(off));
    }
};

struct SampleFile : Samples {                                           // src/samples.rs:44-94
    int fd, format; uint64_t file_len, rate;
    SampleFile(const std::string &fn, int fmt, uint64_t sr) : format(fmt), rate(sr) {
        fd = open(fn.c_str(), O_RDONLY);
        if (fd < 0) bail(std::string(strerror(errno)) + " (os error " + std::to_string(errno) + ")");
        struct stat st; fstat(fd, &st); file_len = (uint64_t)st.st_size;
    }
    ~SampleFile() override { if (fd >= 0) close(fd); }
    uint64_t len() const override { return file_len / qd_pair_bytes(format); }
    uint64_t sample_rate() const override { return rate; }
    size_t read_at(uint64_t off, qd_c32 *into, size_t n) const override {
        if (!(off < len())) bail("assertion failed: off < self.len()");  // :74
        const uint64_t pb = qd_pair_bytes(format);
        std::vector<uint8_t> buf(pb * n);
        ssize_t got = pread(fd, buf.data(), buf.size(), (off_t)(off * pb));
        if (got < 0) bail("read");
        size_t bytes = (size_t)got - (size_t)got % pb;                   // :84
        size_t pairs = bytes / pb;
        if (pairs) qd_check(qd_unpack(format, buf.data(), pairs, into, QD_MEM_HOST), "unpack");
        return pairs;
    }
};

struct Gen : Samples {                                                  // src/gen.rs:16-52
    std::vector<int64_t> cos; uint64_t rate; double seconds;
    Gen(std::vector<int64_t> c, uint64_t sr, double s) : cos(std::move(c)), rate(sr), seconds(s) {
        if (cos.empty()) bail("cos cannot be empty");
        if (rate == 0) bail("sample rate may not be zero");
        if (!(seconds > 0.0)) bail("seconds may not be <= 0");
    }
    uint64_t len() const override {
        double v = seconds * (double)rate;
        return !(v > 0) ? 0 : (v >= 18446744073709551616.0 ? UINT64_MAX : (uint64_t)v);
    }
    uint64_t sample_rate() const override { return rate; }
    size_t read_at(uint64_t off, qd_c32 *buf, size_t n) const override {
        if (n) qd_check(qd_gen(cos.data(), cos.size(), rate, off, n, buf, QD_MEM_HOST), "gen");
        return n;
    }
};

struct Shift : Samples {                                                // src/shift.rs
    std::unique_ptr<Samples> inner; double ratio; uint64_t rate;
    Shift(std::unique_ptr<Samples> in, int64_t freq) : inner(std::move(in)) {
        rate = inner->sample_rate();
        int64_t af = freq < 0 ? -freq : freq;
        if (!(af < (int64_t)(rate / 2))) bail("frequency must be under half the sample rate");
        ratio = qd_shift_ratio(freq, rate);
    }
    uint64_t len() const override { return inner->len(); }
    uint64_t sample_rate() const override { return rate; }
    size_t read_at(uint64_t off, qd_c32 *buf, size_t n) const override {
        size_t valid = inner->read_at(off, buf, n);
        if (valid) qd_check(qd_shift(buf, valid, off, ratio, QD_MEM_HOST), "shift");
        return valid;
    }
};

struct LowPass : Samples {                                              // src/filter.rs
    std::unique_ptr<Samples> inner; std::vector<float> taps; uint64_t decimate, orig_rate;
    LowPass(std::unique_ptr<Samples> in, uint64_t freq, uint64_t dec, size_t size) : inner(std::move(in)), taps(size), decimate(dec) {
        orig_rate = inner->sample_rate();
        qd_check(qd_lowpass_design(freq, orig_rate, size, taps.data()), "lowpass_design");
    }
    uint64_t len() const override {
        if (inner->len() < taps.size()) bail("assertion failed: self.inner.len() >= self.filter.len()");
        return 1 + (inner->len() - taps.size()) / decimate;
    }
    uint64_t sample_rate() const override { return orig_rate / decimate; }
    size_t read_at(uint64_t off, qd_c32 *buf, size_t n) const override {
        std::vector<qd_c32> raw(n * decimate + taps.size());             // :68-69
        size_t valid = inner->read_at(off * decimate, raw.data(), raw.size());
        size_t produced = 0;
        qd_check(qd_lowpass_block(taps.data(), taps.size(), decimate, raw.data(), valid, buf, n, &produced, QD_MEM_HOST), "lowpass");
        return produced;
    }
};

// ------------------------------------------------------------------ sinks

struct ChainSpec {          // what a fused plan can express
    bool fusable = false;
    const Op *src = nullptr, *shift = nullptr, *lowpass = nullptr;
};

// The whole source file for a fused plan: mapped, not read — and registered with the HIP runtime when it allows it, so the
// engine copies straight out of the page cache (QD_MEM_HOST_PINNED) instead of staging every chunk through a pinned ring.
struct MappedFile {
    const uint8_t *p = nullptr;
    size_t size = 0;
    int mem = QD_MEM_HOST;
    explicit MappedFile(const std::string &fn) {
        int fd = open(fn.c_str(), O_RDONLY);
        if (fd < 0) bail(std::string(strerror(errno)) + ": " + fn);
        struct stat st;
        if (fstat(fd, &st) != 0) { close(fd); bail(std::string(strerror(errno)) + ": " + fn); }
        size = (size_t)st.st_size;
        if (size) {
            void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { close(fd); bail(std::string(strerror(errno)) + ": " + fn); }
            p = static_cast<const uint8_t *>(m);
            if (size >= (64u << 20) && qd_host_register(m, size) == QD_OK) mem = QD_MEM_HOST_PINNED;   // small files: not worth pinning
        }
        close(fd);
    }
    ~MappedFile() {
        if (p) {
            if (mem == QD_MEM_HOST_PINNED) (void)qd_host_unregister(const_cast<uint8_t *>(p));
            munmap(const_cast<uint8_t *>(p), size);
        }
    }
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
};

int g_gpus = 1;      // `-gpus N` in front of the chain: shard the sink's windows over N devices (qd_plan_run_sharded)

// plan for the CLI: the library's defaults, plus window-range shards over g_gpus devices (repeating devices when the
// machine has fewer: the shards then run as independent streams of one device)
int create_plan(const qd_chain_desc &d, qd_plan **plan) {
    if (g_gpus <= 1) return qd_plan_create(&d, plan);
    qd_plan_options o{};
    o.struct_size = sizeof o;
    int n_dev = 1;
    if (qd_device_count(&n_dev) != QD_OK || n_dev < 1) n_dev = 1;
    o.n_shards = (uint32_t)(g_gpus > QD_MAX_SHARDS ? QD_MAX_SHARDS : g_gpus);
    for (uint32_t g = 0; g < o.n_shards; ++g) o.shard_device[g] = (int32_t)(g % (uint32_t)n_dev);
    return qd_plan_create_ex(&d, &o, plan);
}

// device buffer that frees itself (the `gen` source of a fused chain lives in HBM)
struct DeviceBuf {
    void *p = nullptr;
    ~DeviceBuf() { if (p) qd_device_free(p); }
};

// sparkfft / bucket through ONE fused plan over the whole file — or over a `gen` stream that is produced on the device
// (src/gen.rs:30-47) and never crosses PCIe: only the glyph codes / digits come back
// returns false when the library has no fused plan for the chain (QD_ERR_UNSUPPORTED: e.g. overlapping windows whose FIR input exceeds one
// workgroup's LDS); the header line is printed by then, the caller pulls the windows through the iterator chain instead
bool run_fused(const ChainSpec &cs, const Op &sink, uint64_t out_rate) {
    const bool from_gen = cs.src->kind == OP_GEN;
    std::unique_ptr<MappedFile> data;
    if (!from_gen) data.reset(new MappedFile(cs.src->filename));
    qd_chain_desc d{};
    d.struct_size = sizeof d;
    d.format = from_gen ? QD_FMT_CF32 : cs.src->format; d.sample_rate = cs.src->sample_rate;
    d.n_samples = from_gen ? (uint64_t)(cs.src->seconds * (double)cs.src->sample_rate)          // Gen::len, src/gen.rs:32
                           : data->size / qd_pair_bytes(cs.src->format);
    if (cs.shift) { d.has_shift = 1; d.shift_hz = cs.shift->shift; }
    if (cs.lowpass) { d.has_lowpass = 1; d.lowpass_hz = cs.lowpass->lp_freq; d.decimate = cs.lowpass->decimate; d.taps = cs.lowpass->size; }
    d.width = sink.width; d.stride = sink.stride;
    d.epilogue = sink.kind == OP_BUCKET ? QD_EPI_BUCKET2_U8 : QD_EPI_GLYPH_U8;
    d.has_range = sink.has_range; d.range_min = sink.rmin; d.range_max = sink.rmax;
    if (sink.kind == OP_SPARKFFT) printf("sparkfft sample_rate=%" PRIu64 "\n", out_rate);   // printed before any read (src/fft.rs:19)
    qd_plan *plan = nullptr;
    {
        const int rc = from_gen ? qd_plan_create(&d, &plan) : create_plan(d, &plan);
        if (rc == QD_ERR_UNSUPPORTED) return false;
        qd_check(rc, "plan");
    }
    qd_plan_info info;
    qd_check(qd_plan_get_info(plan, &info), "plan info");
    std::vector<uint8_t> out(info.n_windows * info.out_bytes_per_window + 1);
    if (info.n_windows && !from_gen) {
        if (g_gpus > 1) qd_check(qd_plan_run_sharded(plan, data->p, data->mem, out.data(), QD_MEM_HOST), "run (sharded)");
        else qd_check(qd_plan_run(plan, data->p, data->mem, 0, d.n_samples, 0, info.n_windows, out.data(), QD_MEM_HOST, nullptr), "run");
    }
    if (info.n_windows && from_gen) {
        DeviceBuf src, dst;
        const size_t ob = (size_t)(info.n_windows * info.out_bytes_per_window);
        qd_check(qd_device_alloc((size_t)d.n_samples * 8, &src.p), "device buffer for gen");
        qd_check(qd_device_alloc(ob, &dst.p), "device buffer for the sink");
        const uint64_t piece = 1ull << 28;                                   // Gen::read_at in pieces: bounded kernel launches
        for (uint64_t a = 0; a < d.n_samples; a += piece) {
            const uint64_t n = d.n_samples - a < piece ? d.n_samples - a : piece;
            qd_check(qd_gen(cs.src->cos.data(), cs.src->cos.size(), cs.src->sample_rate, a, (size_t)n,
                            static_cast<qd_c32 *>(src.p) + a, QD_MEM_DEVICE), "gen");
        }
        qd_check(qd_plan_run(plan, src.p, QD_MEM_DEVICE, 0, d.n_samples, 0, info.n_windows, dst.p, QD_MEM_DEVICE, nullptr), "run");
        qd_check(qd_device_copy(out.data(), QD_MEM_HOST, dst.p, QD_MEM_DEVICE, ob), "copy back");   // synchronises with the launch
    }
    qd_plan_destroy(plan);
    if (sink.kind == OP_SPARKFFT) {
        // header already printed; rows only
        std::string line;
        for (uint64_t w = 0; w < info.n_windows; ++w) {
            line.assign("\xE2\x94\x82");
            for (size_t b = 0; b < sink.width; ++b) {
                uint8_t c = out[w * sink.width + b];
                if (c == 0) line.push_back(' ');
                else if (c <= 8) { line.push_back((char)0xE2); line.push_back((char)0x96); line.push_back((char)(0x80 + c)); }
                else bail("index out of bounds: the len is 7 but the index is 7");
            }
            line += "\xE2\x94\x82\n";
            fwrite(line.data(), 1, line.size(), stdout);
        }
    } else {
        std::string digits;
        for (uint64_t w = 0; w < info.n_windows; ++w) digits.push_back((char)('0' + out[w]));
        printf("%s\n", digits.c_str());                                   // src/lib.rs:144-158
    }
    return true;
}

// FFT + epilogue of nb gathered windows (contiguous, stride W) on the GPU: a no-shift/no-lowpass plan
std::vector<uint8_t> sink_batch(const qd_c32 *buf, uint64_t nb, size_t W, const Op &sink) {
    qd_chain_desc d{};
    d.struct_size = sizeof d;
    d.format = QD_FMT_CF32; d.sample_rate = 1;
    // len such that the sink's own loop yields exactly nb windows at stride W
    d.n_samples = sink.kind == OP_BUCKET ? (nb + 1) * W : nb * W + 1;
    d.width = W; d.stride = W;
    d.epilogue = sink.kind == OP_BUCKET ? QD_EPI_BUCKET2_U8 : QD_EPI_GLYPH_U8;
    d.has_range = sink.has_range; d.range_min = sink.rmin; d.range_max = sink.rmax;
    qd_plan *plan = nullptr;
    qd_check(qd_plan_create(&d, &plan), "plan");
    qd_plan_info info;
    qd_check(qd_plan_get_info(plan, &info), "plan info");
    std::vector<uint8_t> out(nb * info.out_bytes_per_window + 1);
    int rc = qd_plan_run(plan, buf, QD_MEM_HOST, 0, nb * W, 0, nb, out.data(), QD_MEM_HOST, nullptr);
    qd_plan_destroy(plan);
    qd_check(rc, "run");
    return out;
}

// the same sinks over an arbitrary iterator chain: windows are pulled through read_exact_at exactly as
// the reference does (src/fft.rs:30,91), gathered, and transformed in batches on the GPU
void run_iter_sink(const Samples &s, const Op &sink, bool header_printed = false) {
    const size_t W = sink.width; const uint64_t S = sink.stride;
    if (sink.kind == OP_SPARKFFT && !header_printed) printf("sparkfft sample_rate=%" PRIu64 "\n", s.sample_rate());
    if (!W || (W & (W - 1))) bail("Radix4 algorithm requires a power-of-two input size");
    if (S == 0) bail("stride 0 never terminates");
    uint64_t len = s.len();
    if (len < W) bail("attempt to subtract with overflow");               // src/fft.rs:28 / :86
    uint64_t lim = len - W;
    uint64_t nwin = sink.kind == OP_BUCKET ? lim / S : (lim == 0 ? 0 : (lim - 1) / S + 1);
    const uint64_t batch = 4096;
    std::vector<qd_c32> buf(batch * W);
    std::string digits;
    for (uint64_t w0 = 0; w0 < nwin; w0 += batch) {
        uint64_t nb = nwin - w0 < batch ? nwin - w0 : batch;
        for (uint64_t i = 0; i < nb; ++i) s.read_exact_at((w0 + i) * S, buf.data() + i * W, W);
        std::vector<uint8_t> out = sink_batch(buf.data(), nb, W, sink);
        if (sink.kind == OP_SPARKFFT) {
            std::string line;
            for (uint64_t w = 0; w < nb; ++w) {
                line.assign("\xE2\x94\x82");
                for (size_t b = 0; b < W; ++b) {
                    uint8_t c = out[w * W + b];
                    if (c == 0) line.push_back(' ');
                    else if (c <= 8) { line.push_back((char)0xE2); line.push_back((char)0x96); line.push_back((char)(0x80 + c)); }
                    else bail("index out of bounds: the len is 7 but the index is 7");
                }
                line += "\xE2\x94\x82\n";
                fwrite(line.data(), 1, line.size(), stdout);
            }
        } else {
            for (uint64_t w = 0; w < nb; ++w) digits.push_back((char)('0' + out[w]));
        }
    }
    if (sink.kind == OP_BUCKET) printf("%s\n", digits.c_str());
}

// do_write (src/lib.rs:178-213).  When the chain is  from [shift] lowpass  the full 0x1000-sample
// read_at blocks come from ONE fused plan (QD_EPI_CF32_BLOCKS); the ragged end of the stream — where
// every read_at has its own `valid` — and any other chain go through the block iterator.
void do_write(const Samples &s, bool overwrite, const std::string &prefix, const ChainSpec *cs) {
    if (prefix == "-") bail("not implemented");
    std::string fn = prefix + ".sr" + std::to_string(s.sample_rate()) + ".cf32";
    int flags = O_WRONLY | (overwrite ? O_CREAT : (O_CREAT | O_EXCL));
    int fd = open(fn.c_str(), flags, 0644);
    if (fd < 0) bail(std::string(strerror(errno)) + " (os error " + std::to_string(errno) + ")");
    uint64_t off = 0, len;
    try { len = s.len(); } catch (...) { close(fd); throw; }
    if (cs && cs->fusable && cs->src->kind == OP_FROM && cs->lowpass && !getenv("QUADRS_HIP_NO_FUSE")) {
        MappedFile data(cs->src->filename);
        qd_chain_desc d{};
        d.struct_size = sizeof d;
        d.format = cs->src->format; d.sample_rate = cs->src->sample_rate;
        d.n_samples = data.size / qd_pair_bytes(cs->src->format);
        if (cs->shift) { d.has_shift = 1; d.shift_hz = cs->shift->shift; }
        d.has_lowpass = 1; d.lowpass_hz = cs->lowpass->lp_freq; d.decimate = cs->lowpass->decimate; d.taps = cs->lowpass->size;
        d.width = 0x1000; d.stride = 0x1000; d.epilogue = QD_EPI_CF32_BLOCKS;
        qd_plan *plan = nullptr;
        int rc = create_plan(d, &plan);
        if (rc == QD_OK) {
            qd_plan_info info;
            qd_check(qd_plan_get_info(plan, &info), "plan info");
            if (info.n_windows) {
                std::vector<qd_c32> out(info.n_windows * 0x1000);
                rc = g_gpus > 1 ? qd_plan_run_sharded(plan, data.p, data.mem, out.data(), QD_MEM_HOST)
                                : qd_plan_run(plan, data.p, data.mem, 0, d.n_samples, 0, info.n_windows, out.data(), QD_MEM_HOST, nullptr);
                if (rc == QD_OK) {
                    if (write(fd, out.data(), out.size() * sizeof(qd_c32)) < 0) { qd_plan_destroy(plan); close(fd); bail("write failed"); }
                    off = info.n_windows * 0x1000;
                }
            }
            qd_plan_destroy(plan);
        }
        if (rc != QD_OK && rc != QD_ERR_UNSUPPORTED) { close(fd); qd_check(rc, "fused write"); }
    }
    std::vector<qd_c32> buf(0x1000);
    while (off < len) {
        size_t rd;
        try { rd = s.read_at(off, buf.data(), buf.size()); } catch (...) { close(fd); throw; }
        if (rd == 0) { close(fd); bail("assertion failed: short read at offset " + std::to_string(off) + " of " + std::to_string(len)); }
        off += rd;
        if (write(fd, buf.data(), rd * sizeof(qd_c32)) < 0) { close(fd); bail("write failed"); }
    }
    close(fd);
}

void usage() {
    fprintf(stderr,
            "usage: quadrs-hip [-gpus N] \\\n"
            "    from [-sr SAMPLE_RATE] [-format cf32|cs8|cu8|cs16] FILENAME.sr32k.cf32 \\\n"
            "   shift [-]FREQUENCY \\\n"
            " lowpass [-power 20] [-decimate 8] FREQUENCY \\\n"
            "sparkfft [-width 128] [-stride =width] [-range MIN:MAX] \\\n"
            "  bucket [-width 128] [-stride =width] -by freq COUNT \\\n"
            "   write [-overwrite no] FILENAME_PREFIX \\\n"
            "     gen [-cos FREQUENCY]* [-len 1 (second)] SAMPLE_RATE \\\n"
            "\n\nFormat for FREQUENCY, SAMPLE_RATE, and other suffixes: 123, 123k, 123M, 123G\n");
}

}  // namespace

int main(int argc, char **argv) {
    std::vector<std::string> args(argv + 1, argv + argc);
    try {
        // engine option in front of the reference's grammar: -gpus N shards the sink's windows over N devices in this process
        if (args.size() >= 2 && args[0] == "-gpus") {
            g_gpus = atoi(args[1].c_str());
            if (g_gpus < 1 || g_gpus > QD_MAX_SHARDS) { usage(); fprintf(stderr, "Error: -gpus takes 1..%d\n", QD_MAX_SHARDS); return 2; }
            args.erase(args.begin(), args.begin() + 2);
        }
        // -parse-only: print what the grammar (src/args.rs) made of the command line, one operation per line, and stop before
        // any file or device is touched (tests: filename -> (sample rate, format) guessing, defaults, SI suffixes)
        bool parse_only = false;
        if (!args.empty() && args[0] == "-parse-only") { parse_only = true; args.erase(args.begin()); }
        if (args.empty()) { usage(); return 2; }
        std::vector<Op> ops;
        try { ops = parse(args); } catch (const Fail &f) { usage(); fprintf(stderr, "Error: %s\n", f.msg.c_str()); return 2; }
        if (parse_only) {
            static const char *fmt_name[] = {"cf32", "cs8", "cu8", "cs16"};
            for (const Op &op : ops) {
                switch (op.kind) {
                case OP_FROM: printf("from file=%s sample_rate=%llu format=%s\n", op.filename.c_str(), (unsigned long long)op.sample_rate, fmt_name[op.format & 3]); break;
                case OP_GEN: printf("gen cos=%zu sample_rate=%llu seconds=%.17g\n", op.cos.size(), (unsigned long long)op.sample_rate, op.seconds); break;
                case OP_SHIFT: printf("shift %lld\n", (long long)op.shift); break;
                case OP_LOWPASS: printf("lowpass frequency=%llu decimate=%llu size=%zu\n", (unsigned long long)op.lp_freq, (unsigned long long)op.decimate, op.size); break;
                case OP_SPARKFFT: printf("sparkfft width=%zu stride=%llu range=%s\n", op.width, (unsigned long long)op.stride, op.has_range ? "yes" : "no"); break;
                case OP_BUCKET: printf("bucket width=%zu stride=%llu levels=%zu\n", op.width, (unsigned long long)op.stride, op.levels); break;
                case OP_WRITE: printf("write prefix=%s overwrite=%d\n", op.prefix.c_str(), op.overwrite ? 1 : 0); break;
                }
            }
            return 0;
        }

        // fold the commands left to right (src/bin/quadrs.rs:48-56)
        std::unique_ptr<Samples> samples;
        ChainSpec cs;
        bool chain_clean = true;        // from [shift] [lowpass] so far, each at most once, in that order
        for (size_t i = 0; i < ops.size(); ++i) {
            const Op &op = ops[i];
            switch (op.kind) {
            case OP_FROM:
                samples.reset(new SampleFile(op.filename, op.format, op.sample_rate));
                cs = ChainSpec{}; cs.src = &op; cs.fusable = true; chain_clean = true;
                break;
            case OP_GEN:
                samples.reset(new Gen(op.cos, op.sample_rate, op.seconds));
                cs = ChainSpec{}; cs.src = &op; cs.fusable = true; chain_clean = true;
                break;
            case OP_SHIFT:
                if (!samples) bail("shift requires an input");
                if (cs.shift || cs.lowpass) chain_clean = false;
                samples.reset(new Shift(std::move(samples), op.shift));
                cs.shift = &op;
                break;
            case OP_LOWPASS:
                if (!samples) bail("lowpass requires an input");
                if (cs.lowpass) chain_clean = false;
                samples.reset(new LowPass(std::move(samples), op.lp_freq, op.decimate, op.size));
                cs.lowpass = &op;
                break;
            case OP_SPARKFFT:
            case OP_BUCKET:
                if (!samples) bail(op.kind == OP_SPARKFFT ? "sparkfft requires an input" : "bucket -by freq requires an input");
                if (op.kind == OP_BUCKET && op.levels != 2) bail("only supporting two levels for now");
                if (cs.fusable && chain_clean && !getenv("QUADRS_HIP_NO_FUSE")) {
                    if (!run_fused(cs, op, samples->sample_rate())) run_iter_sink(*samples, op, true);
                } else run_iter_sink(*samples, op);
                break;
            case OP_WRITE:
                if (!samples) bail("write requires an input");
                do_write(*samples, op.overwrite, op.prefix, chain_clean ? &cs : nullptr);
                break;
            }
        }
        return 0;
    } catch (const Fail &f) {
        fflush(stdout);
        fprintf(stderr, "Error: %s\n", f.msg.c_str());
        return 1;
    }
}
