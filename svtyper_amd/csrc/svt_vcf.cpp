// svt_vcf.cpp -- bulk VCF body parse + output-line emission (include/svtyper_vcf.h).
//
// Host-only.  The steps either side of the device path for a whole chunk of variant lines: the reference makes a
// `Variant` object, a breakpoint dict and a joined string per line (svtyper/parsers.py:256-399, :125-223;
// svtyper/classic.py:219-278; svtyper/singlesample.py:577-652); at a few hundred thousand sites per second that is
// the whole run.  Here a block of text becomes breakpoint arrays (phase 1: lines in parallel; phase 2: BND pairing
// in order) and, once the device has the results, the output lines of the block as one text.  Every rule below
// names the Python expression it follows; whatever is not expressed EXACTLY (a number Python's int()/float() might
// read differently, a missing key, sample columns carrying FORMAT values, trailing whitespace ...) goes back to the
// caller's per-line Python code, which is also the checker of this file (tests/test_bulk_vcf.py).

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/svtyper_vcf.h"
#include "svt_error.h"
#include "svt_format_fields.h"
#include "svt_host_cpus.h"

namespace {

using svt::fail;
using svt::guarded;
using svt::run_threads;

enum : uint8_t { kDel = 0, kDup = 1, kInv = 2, kBnd = 3 };          // svt_unit.svtype (classic.py:228)
enum : int { K_SVTYPE, K_END, K_CIPOS, K_CIEND, K_CIPOS95, K_CIEND95, K_MATEID, K_N };
const char* const kSpecial[K_N] = {"SVTYPE", "END", "CIPOS", "CIEND", "CIPOS95", "CIEND95", "MATEID"};

// phase-1 classification of a line (LINE_STOP: a BND -- or possibly-BND -- line outside the fast route)
enum : uint8_t { LINE_SITE, LINE_BND, LINE_PYTHON, LINE_SKIPPED, LINE_STOP };

struct Span { uint32_t off = 0, len = 0; };

struct InfoItem {
    const char* key;
    const char* val;      // after the first '=' up to the second (value.index('='), parsers.py:264-266)
    uint32_t klen, vlen;
    bool has_eq;
};

// what to print of a line's INFO items and where its special keys are, for one sequence of keys (the counterpart of
// vcf.Vcf.info_plans: lines of one caller repeat the same key sequence)
struct Plan {
    std::vector<std::string> keys;
    std::vector<uint16_t> order;      // item indices in header declaration order (parsers.py:346-355)
    std::vector<uint8_t> flag;        // parallel: declared Flag -> printed bare
    int special[K_N];
};

struct LineRec {
    uint8_t kind = LINE_PYTHON, svtype = 0, reverse = 0;
    uint32_t chrom = 0;               // index into the thread's chromosome table
    int64_t pos = 0, end = 0;
    int64_t ci[4] = {0, 0, 0, 0};     // CIPOS, CIEND (BND: its own CIPOS in ci[0..1])
    double qual = 0.0;
    Span prefix, suffix;              // "chrom\tpos\tid\tref\talt" and "filter\tinfo" in the thread's arena
    Span id, mate;                    // BND: ID column and MATEID
};

struct ThreadOut {
    std::string arena;
    std::vector<std::string> chroms;
    std::vector<LineRec> lines;
    std::vector<Plan> plans;          // most recently used first
};

struct Held {                         // first mate of a BND pair (Vcf._bnd_pending, parsers.py:157-165)
    bool alive = false;
    std::string id, line, prefix, suffix;
    int32_t chrom = 0;
    int64_t pos = 0, ci[2] = {0, 0};
    uint8_t reverse = 0;
    double qual = 0.0;
};

}  // namespace

struct svt_vcf_parser {
    std::vector<std::string> info_ids;
    std::vector<uint8_t> info_flag;
    std::unordered_map<std::string, uint32_t> info_rank;
    double max_ci_dist = 1e10;
    uint32_t flags = 0;
    std::vector<std::string> chroms;
    std::unordered_map<std::string, int32_t> chrom_index;
    std::vector<Held> held;                                   // insertion order, tombstones
    std::unordered_map<std::string, size_t> held_index;
    std::vector<const Held*> pending_view;
};

struct svt_vcf_chunk {
    std::vector<uint8_t> line_kind;
    std::vector<uint64_t> line_begin;
    std::vector<uint32_t> line_site;
    std::vector<int32_t> chrom_a, chrom_b;
    std::vector<int64_t> pos_a, pos_b, ci, var_length;
    std::vector<uint8_t> svtype, strands;
    std::vector<double> qual_in;
    // text of a site's lines: spans into `arenas` (the threads' arenas, then one for held first mates)
    struct Text { uint32_t arena; Span prefix, suffix; };
    std::vector<Text> first, second;                         // second.arena == UINT32_MAX: no second line
    std::vector<std::string> arenas;
};

namespace {

inline bool plain_int(const char* b, const char* e, int max_digits, int64_t& out)
{
    // the subset of what Python's int() accepts that every reader agrees on: -?[0-9]{1,max_digits}
    bool neg = false;
    if (b < e && *b == '-') { neg = true; ++b; }
    if (b >= e || e - b > max_digits) return false;
    int64_t v = 0;
    for (const char* p = b; p < e; ++p) {
        if (*p < '0' || *p > '9') return false;
        v = v * 10 + (*p - '0');
    }
    out = neg ? -v : v;
    return true;
}

inline bool plain_float(const char* b, const char* e, double& out)
{
    // digits with an optional sign, point and exponent, read by strtod in full (float() and strtod both round
    // correctly); "nan", "inf", hex, underscores, blanks: not here
    if (b >= e || e - b > 40) return false;
    bool digit = false;
    for (const char* p = b; p < e; ++p) {
        const char c = *p;
        if (c >= '0' && c <= '9') digit = true;
        else if (c != '.' && c != 'e' && c != 'E' && c != '+' && c != '-') return false;
    }
    if (!digit) return false;
    char buf[48];
    std::memcpy(buf, b, (size_t)(e - b));
    buf[e - b] = 0;
    char* stop = nullptr;
    const double v = std::strtod(buf, &stop);
    if (stop != buf + (e - b) || !std::isfinite(v)) return false;
    out = v;
    return true;
}

inline bool int_pair(const InfoItem& it, int64_t out[2])
{
    // [int(x) for x in value.split(",")] with exactly two plain elements (parsers.py:12-15)
    if (!it.has_eq) return false;
    const char* b = it.val;
    const char* e = it.val + it.vlen;
    const char* comma = static_cast<const char*>(std::memchr(b, ',', (size_t)(e - b)));
    if (!comma || std::memchr(comma + 1, ',', (size_t)(e - comma - 1))) return false;
    return plain_int(b, comma, 15, out[0]) && plain_int(comma + 1, e, 15, out[1]);
}

inline bool same_keys(const Plan& p, const InfoItem* items, size_t n)
{
    if (p.keys.size() != n) return false;
    for (size_t i = 0; i < n; ++i)
        if (p.keys[i].size() != items[i].klen || std::memcmp(p.keys[i].data(), items[i].key, items[i].klen) != 0) return false;
    return true;
}

const Plan& plan_for(const svt_vcf_parser& P, ThreadOut& T, const InfoItem* items, size_t n)
{
    for (size_t k = 0; k < T.plans.size(); ++k)
        if (same_keys(T.plans[k], items, n)) {
            if (k) std::swap(T.plans[k], T.plans[0]);
            return T.plans[0];
        }
    Plan p;
    for (int s = 0; s < K_N; ++s) p.special[s] = -1;
    std::vector<int> last_of_rank(P.info_ids.size(), -1);
    for (size_t i = 0; i < n; ++i) {
        p.keys.emplace_back(items[i].key, items[i].klen);
        for (int s = 0; s < K_N; ++s)
            if (p.keys.back() == kSpecial[s]) p.special[s] = (int)i;          // a repeated key: the last one wins (dict)
        auto it = P.info_rank.find(p.keys.back());
        if (it != P.info_rank.end()) last_of_rank[it->second] = (int)i;
    }
    for (size_t r = 0; r < last_of_rank.size(); ++r)
        if (last_of_rank[r] >= 0) {
            p.order.push_back((uint16_t)last_of_rank[r]);
            p.flag.push_back(P.info_flag[r]);
        }
    if (T.plans.size() >= 8) T.plans.pop_back();
    T.plans.insert(T.plans.begin(), std::move(p));
    return T.plans[0];
}

inline Span put(std::string& arena, const char* b, size_t n)
{
    Span s;
    s.off = (uint32_t)arena.size();
    s.len = (uint32_t)n;
    arena.append(b, n);
    return s;
}

inline void put_i64(std::string& s, int64_t v)
{
    char buf[24];
    char* p = buf + sizeof buf;
    uint64_t u = v < 0 ? 0ull - (uint64_t)v : (uint64_t)v;
    do { *--p = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *--p = '-';
    s.append(p, (size_t)(buf + sizeof buf - p));
}

uint32_t intern_local(ThreadOut& T, const char* b, size_t n)
{
    for (size_t k = T.chroms.size(); k-- > 0;)      // (a sorted VCF: the last one, nearly always)
        if (T.chroms[k].size() == n && std::memcmp(T.chroms[k].data(), b, n) == 0) return (uint32_t)k;
    T.chroms.emplace_back(b, n);
    return (uint32_t)T.chroms.size() - 1;
}

// One line (no newline) -> LineRec.  Everything Variant.__init__ + get_variant_breakpoints would do with it, or
// LINE_PYTHON / LINE_STOP when the general code has to.
void parse_line(const svt_vcf_parser& P, ThreadOut& T, const char* b, const char* e, LineRec& r)
{
    r = LineRec();
    if (b < e && *b == '#' && (P.flags & SVT_VCF_SKIP_HASH_LINES)) { r.kind = LINE_SKIPPED; return; }   // singlesample.py:589
    if (b == e) { r.kind = LINE_PYTHON; return; }                        // Variant(['']): fewer than 8 columns, the caller exits
    // line.rstrip(): a line that ends in a printable ASCII character loses nothing; anything else is left to str.rstrip
    if ((unsigned char)e[-1] < 0x21 || (unsigned char)e[-1] > 0x7e) { r.kind = LINE_STOP; return; }
    const char* col[11];
    int ncol = 0;
    col[0] = b;
    for (const char* p = b; ncol < 10;) {
        const char* t = static_cast<const char*>(std::memchr(p, '\t', (size_t)(e - p)));
        if (!t) break;
        col[++ncol] = t + 1;
        p = t + 1;
    }
    ++ncol;                                                               // columns seen (at most 11: the rest is not needed)
    if (ncol < 8) { r.kind = LINE_PYTHON; return; }                       // "VCF file must have at least 8 columns": exit(1)
    auto col_end = [&](int k) { return k + 1 < ncol ? col[k + 1] - 1 : e; };
    bool odd = false;
    // a line with sample columns is only taken when they carry nothing but a GT (parsers.py:296-307; the columns this
    // run writes replace them: pipeline.SampleColumnWriter.eligible)
    if (ncol >= 10 && !(col_end(8) - col[8] == 2 && col[8][0] == 'G' && col[8][1] == 'T')) odd = true;
    if (!plain_int(col[1], col_end(1), 15, r.pos)) odd = true;            // int(var_list[1])
    if (col_end(5) - col[5] == 1 && col[5][0] == '.') r.qual = 0.0;        // parsers.py:285-288
    else if (!plain_float(col[5], col_end(5), r.qual)) odd = true;
    if (!(P.flags & SVT_VCF_SUM_QUALS)) r.qual = 0.0;                     // classic.py:227-228
    // INFO: item.split('=') -- key, and what lies between the first and the second '=' (parsers.py:260-268)
    InfoItem items[256];
    size_t n_items = 0;
    {
        const char* p = col[7];
        const char* const ie = col_end(7);
        for (;;) {
            const char* semi = static_cast<const char*>(std::memchr(p, ';', (size_t)(ie - p)));
            const char* item_end = semi ? semi : ie;
            if (n_items == 256) { r.kind = LINE_STOP; return; }           // (not worth a heap path: the general code takes it)
            InfoItem& it = items[n_items++];
            const char* eq = static_cast<const char*>(std::memchr(p, '=', (size_t)(item_end - p)));
            it.key = p;
            it.has_eq = eq != nullptr;
            if (eq) {
                it.klen = (uint32_t)(eq - p);
                const char* eq2 = static_cast<const char*>(std::memchr(eq + 1, '=', (size_t)(item_end - eq - 1)));
                it.val = eq + 1;
                it.vlen = (uint32_t)((eq2 ? eq2 : item_end) - eq - 1);
            } else {
                it.klen = (uint32_t)(item_end - p);
                it.val = nullptr;
                it.vlen = 0;
            }
            if (!semi) break;
            p = semi + 1;
        }
    }
    const Plan& plan = plan_for(P, T, items, n_items);
    // SVTYPE decides who may handle the line: a BND line off the fast route stops the bulk parse (pairing is stateful)
    const int sv = plan.special[K_SVTYPE];
    int svtype = -1;
    if (sv >= 0 && items[sv].has_eq) {
        const InfoItem& it = items[sv];
        if (it.vlen == 3) {
            if (!std::memcmp(it.val, "DEL", 3)) svtype = kDel;
            else if (!std::memcmp(it.val, "DUP", 3)) svtype = kDup;
            else if (!std::memcmp(it.val, "INV", 3)) svtype = kInv;
            else if (!std::memcmp(it.val, "BND", 3)) svtype = kBnd;
        }
    }
    if (svtype < 0) { r.kind = LINE_PYTHON; return; }                     // missing / unsupported SVTYPE: a warning and the raw line
    r.svtype = (uint8_t)svtype;
    auto interval = [&](int key, int key95, int64_t out[2]) {              // confidence_interval(), parsers.py:11-15
        if (plan.special[key] < 0 || !int_pair(items[plan.special[key]], out)) return false;
        if ((double)(out[1] - out[0]) > P.max_ci_dist)
            return plan.special[key95] >= 0 && int_pair(items[plan.special[key95]], out);
        return true;
    };
    if (svtype == kBnd) {
        const int m = plan.special[K_MATEID];
        if (m < 0 || !items[m].has_eq) odd = true;
        if (!interval(K_CIPOS, K_CIPOS95, r.ci)) odd = true;
        if (col_end(4) == col[4]) odd = true;                             // alt[-1]
        if (odd) { r.kind = LINE_STOP; return; }
        const char last = col_end(4)[-1];
        r.reverse = (last != '[' && last != ']') ? 1 : 0;                 // parsers.py:170,176
        r.id = put(T.arena, col[2], (size_t)(col_end(2) - col[2]));
        r.mate = put(T.arena, items[m].val, items[m].vlen);
        r.kind = LINE_BND;
    } else {
        const int en = plan.special[K_END];
        if (en < 0 || !items[en].has_eq || !plain_int(items[en].val, items[en].val + items[en].vlen, 15, r.end)) odd = true;
        if (!interval(K_CIPOS, K_CIPOS95, r.ci)) odd = true;
        if (!interval(K_CIEND, K_CIEND95, r.ci + 2)) odd = true;
        if (odd) { r.kind = LINE_PYTHON; return; }
        r.kind = LINE_SITE;
    }
    r.chrom = intern_local(T, col[0], (size_t)(col_end(0) - col[0]));
    // the eight fixed columns as get_var_string prints them (parsers.py:357-373): POS through int(), INFO in header order
    std::string& a = T.arena;
    r.prefix.off = (uint32_t)a.size();
    a.append(col[0], (size_t)(col_end(0) - col[0]));
    a += '\t';
    put_i64(a, r.pos);
    a.append(col[2] - 1, (size_t)(col_end(4) - (col[2] - 1)));            // "\tid\tref\talt" as it stands
    r.prefix.len = (uint32_t)a.size() - r.prefix.off;
    r.suffix.off = (uint32_t)a.size();
    a.append(col[6], (size_t)(col_end(6) - col[6]));
    a += '\t';
    for (size_t k = 0; k < plan.order.size(); ++k) {
        const InfoItem& it = items[plan.order[k]];
        if (k) a += ';';
        a.append(it.key, it.klen);
        if (plan.flag[k]) continue;
        a += '=';
        if (it.has_eq) a.append(it.val, it.vlen);
        else a += "True";                                                 // '%s=%s' % (key, True)
    }
    r.suffix.len = (uint32_t)a.size() - r.suffix.off;
}

int32_t intern_global(svt_vcf_parser& P, const std::string& name)
{
    auto it = P.chrom_index.find(name);
    if (it != P.chrom_index.end()) return it->second;
    const int32_t k = (int32_t)P.chroms.size();
    P.chroms.push_back(name);
    P.chrom_index.emplace(name, k);
    return k;
}

int parse_impl(svt_vcf_parser* P, const char* text, size_t len, svt_vcf_chunk** out, size_t* consumed)
{
    if (!P || !out || !consumed || (len && !text)) return fail(SVT_ERR_INVALID, "null argument");
    if (len >= (1ull << 32)) return fail(SVT_ERR_INVALID, "svt_vcf_parse: blocks of at most 4 GiB");
    *out = nullptr;
    *consumed = 0;
    const bool trace = std::getenv("SVT_TRACE_VCF") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    std::unique_ptr<svt_vcf_chunk> C(new svt_vcf_chunk());
    std::vector<uint64_t>& lb = C->line_begin;
    for (size_t at = 0; at < len;) {
        lb.push_back(at);
        const char* nl = static_cast<const char*>(std::memchr(text + at, '\n', len - at));
        at = nl ? (size_t)(nl - text) + 1 : len;
    }
    const size_t n_lines = lb.size();
    lb.push_back(len);
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(svt::usable_cpus(), n_lines / 2048 + 1));
    std::vector<ThreadOut> T(nt);
    const double t1 = now();
    run_threads(nt, [&](unsigned t) {
        const size_t lo = n_lines * t / nt, hi = n_lines * (t + 1) / nt;
        ThreadOut o;                           // (a local: neighbours in T share cache lines, and every append moves a size field)
        o.lines.resize(hi - lo);
        o.arena.reserve((size_t)(lb[hi] - lb[lo]));
        for (size_t i = lo; i < hi; ++i) {
            const char* b = text + lb[i];
            const char* e = text + lb[i + 1];
            if (e > b && e[-1] == '\n') --e;
            parse_line(*P, o, b, e, o.lines[i - lo]);
        }
        T[t] = std::move(o);
    });
    // phase 2, in line order: chromosome indices, BND pairing (parsers.py:157-185), the site arrays
    const double t2 = now();
    C->arenas.resize(nt + 1);
    C->line_kind.reserve(n_lines);
    C->line_site.reserve(n_lines);
    for (auto* v : {&C->chrom_a, &C->chrom_b}) v->reserve(n_lines);
    for (auto* v : {&C->pos_a, &C->pos_b, &C->var_length}) v->reserve(n_lines);
    C->ci.reserve(n_lines * 4);
    for (auto* v : {&C->svtype, &C->strands}) v->reserve(n_lines);
    C->qual_in.reserve(n_lines);
    C->first.reserve(n_lines);
    C->second.reserve(n_lines);
    std::string& held_arena = C->arenas[nt];
    bool stopped = false;
    size_t done = 0;
    for (unsigned t = 0; t < nt && !stopped; ++t) {
        ThreadOut& o = T[t];
        std::vector<int32_t> chrom_map(o.chroms.size(), -1);
        const size_t lo = n_lines * t / nt;
        for (size_t k = 0; k < o.lines.size(); ++k) {
            const LineRec& r = o.lines[k];
            if (r.kind == LINE_STOP) { stopped = true; break; }
            done = lo + k + 1;
            if (r.kind == LINE_PYTHON || r.kind == LINE_SKIPPED) {
                C->line_kind.push_back(r.kind == LINE_PYTHON ? SVT_VCF_LINE_PYTHON : SVT_VCF_LINE_SKIPPED);
                C->line_site.push_back(0);
                continue;
            }
            if (chrom_map[r.chrom] < 0) chrom_map[r.chrom] = intern_global(*P, o.chroms[r.chrom]);
            const int32_t chrom = chrom_map[r.chrom];
            svt_vcf_chunk::Text mine{t, r.prefix, r.suffix}, none{UINT32_MAX, Span(), Span()};
            if (r.kind == LINE_BND) {
                const std::string mate_id(o.arena.data() + r.mate.off, r.mate.len);
                auto it = P->held_index.find(mate_id);
                if (it == P->held_index.end()) {                          // first of its pair: keep it (parsers.py:162-165)
                    std::string id(o.arena.data() + r.id.off, r.id.len);
                    Held h;
                    h.alive = true;
                    h.id = id;
                    const char* lbeg = text + lb[lo + k];
                    const char* lend = text + lb[lo + k + 1];
                    if (lend > lbeg && lend[-1] == '\n') --lend;
                    h.line.assign(lbeg, lend);
                    h.prefix.assign(o.arena.data() + r.prefix.off, r.prefix.len);
                    h.suffix.assign(o.arena.data() + r.suffix.off, r.suffix.len);
                    h.chrom = chrom;
                    h.pos = r.pos;
                    h.ci[0] = r.ci[0];
                    h.ci[1] = r.ci[1];
                    h.reverse = r.reverse;
                    h.qual = r.qual;
                    auto again = P->held_index.find(id);
                    if (again != P->held_index.end()) P->held[again->second] = std::move(h);   // same ID held twice: replaced in place
                    else {
                        P->held_index.emplace(id, P->held.size());
                        P->held.push_back(std::move(h));
                    }
                    C->line_kind.push_back(SVT_VCF_LINE_HELD);
                    C->line_site.push_back(0);
                    continue;
                }
                Held& f = P->held[it->second];
                svt_vcf_chunk::Text first{nt, put(held_arena, f.prefix.data(), f.prefix.size()), Span()};
                first.suffix = put(held_arena, f.suffix.data(), f.suffix.size());
                C->line_kind.push_back(SVT_VCF_LINE_SITE);
                C->line_site.push_back((uint32_t)C->svtype.size());
                C->chrom_a.push_back(f.chrom);
                C->chrom_b.push_back(chrom);
                C->pos_a.push_back(f.pos + (f.reverse ? 1 : 0));          // parsers.py:219-222
                C->pos_b.push_back(r.pos + (r.reverse ? 1 : 0));
                C->ci.insert(C->ci.end(), {f.ci[0], f.ci[1], r.ci[0], r.ci[1]});
                C->var_length.push_back(0);
                C->svtype.push_back(kBnd);
                C->strands.push_back((uint8_t)(f.reverse | (r.reverse << 1)));
                C->qual_in.push_back(f.qual);                             // the pair is written with the first mate's QUAL
                C->first.push_back(first);
                C->second.push_back(mine);
                f = Held();                                               // del _bnd_pending[first.var_id]
                P->held_index.erase(it);
                continue;
            }
            // DEL / DUP / INV: strands (parsers.py:189-203), var_length of a DEL before the shift (:204-205)
            const uint8_t rev = r.svtype == kDel ? 2 : r.svtype == kDup ? 1 : 0;
            C->line_kind.push_back(SVT_VCF_LINE_SITE);
            C->line_site.push_back((uint32_t)C->svtype.size());
            C->chrom_a.push_back(chrom);
            C->chrom_b.push_back(chrom);
            C->pos_a.push_back(r.pos + (rev & 1));
            C->pos_b.push_back(r.end + (rev >> 1));
            C->ci.insert(C->ci.end(), {r.ci[0], r.ci[1], r.ci[2], r.ci[3]});
            C->var_length.push_back(r.svtype == kDel ? r.end - r.pos : 0);
            C->svtype.push_back(r.svtype);
            C->strands.push_back(rev);
            C->qual_in.push_back(r.qual);
            C->first.push_back(mine);
            C->second.push_back(none);
        }
    }
    for (unsigned t = 0; t < nt; ++t) C->arenas[t] = std::move(T[t].arena);
    C->line_begin.resize(done + 1);
    if (P->held.size() > 1024 && P->held_index.size() * 2 < P->held.size()) {   // drop the tombstones
        std::vector<Held> live;
        for (Held& h : P->held)
            if (h.alive) live.push_back(std::move(h));
        P->held.swap(live);
        P->held_index.clear();
        for (size_t k = 0; k < P->held.size(); ++k) P->held_index.emplace(P->held[k].id, k);
    }
    P->pending_view.clear();
    if (trace) std::fprintf(stderr, "[svt_vcf_parse] %zu lines, %u threads: line scan %.2f ms, lines %.2f ms, pairing + arrays %.2f ms\n", n_lines, nt, t1 - t0, t2 - t1, now() - t2);
    *consumed = (size_t)C->line_begin[done];
    *out = C.release();
    return SVT_OK;
}

inline void put_text(std::string& s, const svt_vcf_chunk& C, const svt_vcf_chunk::Text& t, double qual)
{
    const std::string& a = C.arenas[t.arena];
    s.append(a.data() + t.prefix.off, t.prefix.len);
    s += '\t';
    char buf[64];
    const int n = svt::format_fixed(buf, qual, 2);                        // '%0.2f' % self.qual
    if (n > 0) s.append(buf, (size_t)n);
    else if (std::isnan(qual)) s += "nan";
    else if (std::isinf(qual)) s += qual < 0 ? "-inf" : "inf";
    else svt::fmt::put_fmt(s, "%0.2f", qual);
    s += '\t';
    s.append(a.data() + t.suffix.off, t.suffix.len);
    s += '\t';
}

int emit_impl(const svt_vcf_chunk* C, const svt_result* res, uint32_t n_samples, int qual_mode, const uint8_t* fields,
              uint32_t n_fields, int skipped_as_dots, const char* format_string, char** text_out, uint64_t** offsets_out)
{
    if (!C || !text_out || !offsets_out || !format_string || (n_fields && !fields)) return fail(SVT_ERR_INVALID, "null argument");
    *text_out = nullptr;
    *offsets_out = nullptr;
    const uint64_t n_sites = C->svtype.size();
    if (n_sites && (!res || n_samples == 0)) return fail(SVT_ERR_INVALID, "svt_vcf_emit: no results for the chunk's sites");
    if (qual_mode != SVT_VCF_QUAL_SSO && qual_mode != SVT_VCF_QUAL_CLASSIC) return fail(SVT_ERR_INVALID, "unknown QUAL mode");
    for (uint32_t k = 0; k < n_fields; ++k)
        if (fields[k] >= SVT_N_FORMAT_FIELDS && fields[k] != SVT_FMT_ABSENT) return fail(SVT_ERR_INVALID, "unknown FORMAT field code");
    const size_t fs_len = std::strlen(format_string);
    const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(svt::usable_cpus(), n_sites * n_samples / 2048 + 1));
    std::vector<std::string> part(nt);
    std::vector<std::vector<uint32_t>> len(nt);
    run_threads(nt, [&](unsigned t) {
        const uint64_t lo = n_sites * t / nt, hi = n_sites * (t + 1) / nt;
        std::string s, cols;                   // (locals, moved out at the end: see parse_impl)
        std::vector<uint32_t> lens;
        s.reserve((size_t)(hi - lo) * (256 + 72 * (size_t)n_samples));
        lens.reserve((size_t)(hi - lo));
        for (uint64_t i = lo; i < hi; ++i) {
            const size_t at = s.size();
            const svt_result* r = res + i * n_samples;
            double qual = C->qual_in[i];
            bool all_skipped = qual_mode == SVT_VCF_QUAL_CLASSIC;
            for (uint32_t k = 0; k < n_samples; ++k) {
                const int gt = r[k].gt;
                if (gt >= 0) qual += r[k].sq;                                              // classic.py:485, singlesample.py:546
                else if (gt == SVT_GT_BLANK && qual_mode == SVT_VCF_QUAL_CLASSIC) qual = 0; // classic.py:498
                if (gt != SVT_GT_SKIPPED) all_skipped = false;
            }
            cols.clear();
            if (all_skipped) {                                            // classic.py:282-284 for every sample: only GT was ever set
                cols = "GT";
                for (uint32_t k = 0; k < n_samples; ++k) cols += "\t./.";
            } else {
                cols.append(format_string, fs_len);
                for (uint32_t k = 0; k < n_samples; ++k) {
                    cols += '\t';
                    for (uint32_t f = 0; f < n_fields; ++f) {
                        if (f) cols += ':';
                        if (fields[f] == SVT_FMT_ABSENT) cols += '.';
                        else svt::fmt::put_field(cols, r[k], fields[f], skipped_as_dots != 0);
                    }
                }
            }
            cols += '\n';
            put_text(s, *C, C->first[i], qual);
            s += cols;
            if (C->second[i].arena != UINT32_MAX) {                       // BND: the second mate, same QUAL and genotypes
                put_text(s, *C, C->second[i], qual);                      //   (classic.py:517-521, singlesample.py:647-652)
                s += cols;
            }
            lens.push_back((uint32_t)(s.size() - at));
        }
        part[t] = std::move(s);
        len[t] = std::move(lens);
    });
    uint64_t total = 0;
    for (const auto& s : part) total += s.size();
    char* text = static_cast<char*>(std::malloc(std::max<uint64_t>(total, 1)));
    uint64_t* off = static_cast<uint64_t*>(std::malloc((n_sites + 1) * sizeof(uint64_t)));
    if (!text || !off) {
        std::free(text);
        std::free(off);
        return fail(SVT_ERR_NOMEM, "out of host memory");
    }
    uint64_t at = 0, u = 0;
    for (unsigned t = 0; t < nt; ++t) {
        std::memcpy(text + at, part[t].data(), part[t].size());
        for (uint32_t l : len[t]) {
            off[u++] = at;
            at += l;
        }
    }
    off[n_sites] = at;
    *text_out = text;
    *offsets_out = off;
    return SVT_OK;
}

}  // namespace

extern "C" {

int svt_vcf_parser_create(const char* const* info_ids, const uint8_t* info_is_flag, uint32_t n_info, double max_ci_dist,
                          uint32_t flags, svt_vcf_parser** out)
{
    return guarded([&] {
        if (!out || (n_info && (!info_ids || !info_is_flag))) return fail(SVT_ERR_INVALID, "null argument");
        *out = nullptr;
        if (n_info > 60000) return fail(SVT_ERR_INVALID, "svt_vcf_parser_create: too many INFO declarations");
        std::unique_ptr<svt_vcf_parser> p(new svt_vcf_parser());
        for (uint32_t k = 0; k < n_info; ++k) {
            if (!info_ids[k]) return fail(SVT_ERR_INVALID, "null INFO id");
            p->info_ids.emplace_back(info_ids[k]);
            p->info_flag.push_back(info_is_flag[k] ? 1 : 0);
            p->info_rank.emplace(p->info_ids.back(), k);                  // (Vcf.add_info keeps the first declaration of an id)
        }
        p->max_ci_dist = max_ci_dist;
        p->flags = flags;
        *out = p.release();
        return (int)SVT_OK;
    });
}

void svt_vcf_parser_free(svt_vcf_parser* p) { delete p; }

uint32_t svt_vcf_parser_n_chroms(const svt_vcf_parser* p) { return p ? (uint32_t)p->chroms.size() : 0; }

const char* svt_vcf_parser_chrom(const svt_vcf_parser* p, uint32_t index)
{
    return p && index < p->chroms.size() ? p->chroms[index].c_str() : nullptr;
}

uint32_t svt_vcf_parser_n_pending(const svt_vcf_parser* p)
{
    if (!p) return 0;
    svt_vcf_parser* q = const_cast<svt_vcf_parser*>(p);
    q->pending_view.clear();
    for (const Held& h : p->held)
        if (h.alive) q->pending_view.push_back(&h);
    return (uint32_t)q->pending_view.size();
}

const char* svt_vcf_parser_pending_line(const svt_vcf_parser* p, uint32_t index)
{
    return p && index < p->pending_view.size() ? p->pending_view[index]->line.c_str() : nullptr;
}

int svt_vcf_parse(svt_vcf_parser* p, const char* text, size_t len, svt_vcf_chunk** out, size_t* consumed)
{
    return guarded([&] { return parse_impl(p, text, len, out, consumed); });
}

void svt_vcf_chunk_free(svt_vcf_chunk* c) { delete c; }

int svt_vcf_chunk_view(const svt_vcf_chunk* c, svt_vcf_view* v)
{
    return guarded([&] {
        if (!c || !v) return fail(SVT_ERR_INVALID, "null argument");
        v->n_lines = c->line_kind.size();
        v->line_kind = c->line_kind.data();
        v->line_begin = c->line_begin.data();
        v->line_site = c->line_site.data();
        v->n_sites = c->svtype.size();
        v->chrom_a = c->chrom_a.data();
        v->chrom_b = c->chrom_b.data();
        v->pos_a = c->pos_a.data();
        v->pos_b = c->pos_b.data();
        v->ci = c->ci.data();
        v->var_length = c->var_length.data();
        v->svtype = c->svtype.data();
        v->strands = c->strands.data();
        v->qual_in = c->qual_in.data();
        return (int)SVT_OK;
    });
}

int svt_vcf_emit(const svt_vcf_chunk* c, const svt_result* results, uint32_t n_samples, int qual_mode, const uint8_t* fields,
                 uint32_t n_fields, int skipped_as_dots, const char* format_string, char** text_out, uint64_t** site_offset_out)
{
    return guarded([&] {
        return emit_impl(c, results, n_samples, qual_mode, fields, n_fields, skipped_as_dots, format_string, text_out, site_offset_out);
    });
}

}  // extern "C"
