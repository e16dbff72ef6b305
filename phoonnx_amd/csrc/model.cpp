// model.cpp — resolve VITS parameters from the exported graph and pack them for gfx950.
//
// Parameters are found by FOLLOWING GRAPH NODES, not by trusting initializer names:
// the exporter folds weight-normed flow convs into anonymous `onnx::Conv_N` tensors,
// folds Neg(logs) of ElementwiseAffine into `onnx::Exp_N`, and de-duplicates identical
// initializers.  Node names carry the module path
// ("/flow/flows.6/enc/in_layers.0/Conv"), which is what we key on.
#include "model.hpp"
#include "g2p_model.hpp"

#include <functional>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <stdexcept>

namespace vitsmi {
namespace {

struct TRef {
    const float *p = nullptr;
    std::vector<int64_t> dims;
    int64_t numel() const {
        int64_t n = 1;
        for (auto d : dims) n *= d;
        return n;
    }
};

struct Resolver {
    std::map<std::string, TRef> t;
    std::map<std::string, int64_t> ints;
    std::vector<std::vector<float>> owned;
    std::string name_warnings;  // resolve(): every node whose name and structural position disagree, one per line

    void put(const std::string &name, const OnnxTensor *ot) {
        if (!ot || !ot->data() || t.count(name)) return;
        // every parameter is finite: with that, a NaN / inf can only enter a run through its inputs, where the engine
        // checks for it (the f16 range guard of the generator relies on this)
        {
            const float *p = ot->data();
            const int64_t n = ot->numel();
            for (int64_t i = 0; i < n; i++)
                if (!std::isfinite(p[i])) throw std::runtime_error("non-finite value in parameter " + name);
        }
        TRef r;
        r.p = ot->data();
        r.dims = ot->dims;
        t[name] = r;
    }
    const TRef *get(const std::string &name) const {
        auto it = t.find(name);
        return it == t.end() ? nullptr : &it->second;
    }
    const TRef &req(const std::string &name) const {
        auto it = t.find(name);
        if (it == t.end()) throw std::runtime_error("parameter not found in graph: " + name);
        return it->second;
    }
    // a required tensor whose shape is checked before anything indexes dims[] or copies `numel` floats out of it:
    // shapes in a downloaded voice file are untrusted input
    const TRef &need(const std::string &name, size_t rank, int64_t want_numel = -1) const {
        const TRef &r = req(name);
        if (r.dims.size() != rank) throw std::runtime_error(name + ": expected rank " + std::to_string(rank));
        for (auto d : r.dims)
            if (d <= 0) throw std::runtime_error(name + ": empty dimension");
        if (want_numel >= 0 && r.numel() != want_numel)
            throw std::runtime_error(name + ": expected " + std::to_string(want_numel) + " values, file has " +
                                     std::to_string(r.numel()));
        return r;
    }
    // optional bias of `n` values
    const TRef *bias_of(const std::string &name, int64_t n) const {
        const TRef *b = get(name);
        if (b && b->numel() != n) throw std::runtime_error(name + ": expected " + std::to_string(n) + " values");
        return b;
    }
    int64_t geti(const std::string &k, int64_t d) const {
        auto it = ints.find(k);
        return it == ints.end() ? d : it->second;
    }
};

void split_path(const std::string &node_name, std::string &mod, std::string &leaf) {
    std::vector<std::string> parts;
    std::string cur;
    for (char c : node_name) {
        if (c == '/') {
            if (!cur.empty()) parts.push_back(cur);
            cur.clear();
        } else
            cur.push_back(c);
    }
    if (!cur.empty()) parts.push_back(cur);
    mod.clear();
    leaf = parts.empty() ? "" : parts.back();
    for (size_t i = 0; i + 1 < parts.size(); i++) {
        if (i) mod += ".";
        mod += parts[i];
    }
}

// ---- structure-keyed naming (SURVEY §7 "Weight lookup", App. B).  Node names carry module paths only in exports of
// recent torch versions; older Piper-era exports name their nodes "Conv_123".  The ORDER of the nodes, though, is the
// order SynthesizerTrn.infer executes its modules in (export_onnx.py:318-327 traces it), and every module is
// recognisable by its position in that order plus the (Cout, Cin, kernel, group) of its convolution.  This walks the
// Conv / ConvTranspose / Gather / LayerNorm Mul-Add / Pad / Sub / Exp nodes in graph order and gives each one the
// module-path name the current exporter would have given it ("/flow/flows.6/enc/in_layers.0/Conv"); resolve() then
// runs on those names.  When the graph has real module-path names the two must agree (cross-check).
// Returns "" on success (paths[i] = synthetic name of node i or ""), else what did not fit.
std::string structural_paths(const OnnxModel &om, std::vector<std::string> &paths) {
    struct CV {
        int node, Cout, Cin, K, group, dil;
        bool T;
    };
    const int nn = int(om.nodes.size());
    paths.assign(size_t(nn), std::string());
    std::vector<CV> cv;
    for (int i = 0; i < nn; i++) {
        const OnnxNode &n = om.nodes[size_t(i)];
        if ((n.op != "Conv" && n.op != "ConvTranspose") || n.inputs.size() < 2) continue;
        const OnnxTensor *w = om.find_init(n.inputs[1]);
        if (!w || w->dtype != 1 || w->dims.size() != 3 || !w->data()) return "a convolution without a rank-3 float weight (" + n.op + ")";
        CV c;
        c.node = i;
        c.T = n.op == "ConvTranspose";
        auto g = n.ints.find("group");
        c.group = (g != n.ints.end() && !g->second.empty()) ? int(g->second[0]) : 1;
        auto dl = n.ints.find("dilations");
        c.dil = (dl != n.ints.end() && !dl->second.empty()) ? int(dl->second[0]) : 1;
        c.K = int(w->dims[2]);
        if (c.T) {
            c.Cin = int(w->dims[0]);
            c.Cout = int(w->dims[1]) * c.group;
        } else {
            c.Cout = int(w->dims[0]);
            c.Cin = int(w->dims[1]) * c.group;
        }
        cv.push_back(c);
    }
    const size_t n = cv.size();
    size_t p = 0;
    auto softmax_between = [&](int a, int b) {
        for (int i = a + 1; i < b; i++)
            if (om.nodes[size_t(i)].op == "Softmax") return true;
        return false;
    };
    auto put = [&](size_t pos, const std::string &mod) { paths[size_t(cv[pos].node)] = "/" + mod + (cv[pos].T ? "/ConvTranspose" : "/Conv"); };
    auto is11 = [&](size_t pos) { return pos < n && !cv[pos].T && cv[pos].K == 1 && cv[pos].group == 1; };
    bool gin = false;
    for (const auto &in : om.inputs) gin = gin || in == "sid";
    const std::string S = std::to_string(0);
    (void)S;
    // ---- text encoder: [conv_q conv_k conv_v (Softmax) conv_o | ffn conv_1 conv_2] per layer, then proj
    int H = 0, layers = 0;
    while (p + 6 < n && is11(p) && is11(p + 1) && is11(p + 2) && is11(p + 3) && cv[p].Cout == cv[p].Cin &&
           cv[p + 1].Cout == cv[p].Cout && cv[p + 2].Cout == cv[p].Cout && cv[p + 1].Cin == cv[p].Cin &&
           softmax_between(cv[p + 2].node, cv[p + 3].node)) {
        const std::string a = "enc_p/encoder/attn_layers." + std::to_string(layers), f = "enc_p/encoder/ffn_layers." + std::to_string(layers);
        H = cv[p].Cout;
        put(p, a + "/conv_q");
        put(p + 1, a + "/conv_k");
        put(p + 2, a + "/conv_v");
        put(p + 3, a + "/conv_o");
        put(p + 4, f + "/conv_1");
        put(p + 5, f + "/conv_2");
        if (cv[p + 4].Cin != H || cv[p + 5].Cout != H || cv[p + 4].Cout != cv[p + 5].Cin) return "encoder layer " + std::to_string(layers) + ": unexpected FFN shape";
        p += 6;
        layers++;
    }
    if (!layers) return "no attention layer found at the head of the graph";
    if (!is11(p) || cv[p].Cin != H || cv[p].Cout % 2) return "enc_p.proj not found after the encoder layers";
    const int C = cv[p].Cout / 2;
    put(p++, "enc_p/proj");
    // ---- duration predictor
    auto dds = [&](const std::string &base) -> bool {  // (convs_sep.i, convs_1x1.i) pairs
        int i = 0;
        while (p + 1 < n && cv[p].group > 1 && cv[p].group == cv[p].Cout && is11(p + 1)) {
            put(p, base + "/convs_sep." + std::to_string(i));
            put(p + 1, base + "/convs_1x1." + std::to_string(i));
            p += 2;
            i++;
        }
        return i > 0;
    };
    const size_t sep_at = p + 1 + (gin ? 1 : 0);
    if (sep_at < n && cv[sep_at].group > 1) {  // stochastic duration predictor (models.py:63-70,108-117)
        if (!is11(p) || cv[p].Cin != H) return "dp.pre not found";
        put(p++, "dp/pre");
        if (gin) put(p++, "dp/cond");
        if (!dds("dp/convs")) return "dp.convs not found";
        if (!is11(p)) return "dp.proj not found";
        put(p++, "dp/proj");
        // ConvFlows in execution order; numbered as models.py:109-110 leaves them: 2 k + 1 for k = count .. 1
        std::vector<size_t> starts;
        {
            size_t q = p;
            while (q < n && is11(q) && cv[q].Cin == 1 && cv[q].Cout > 1) {
                starts.push_back(q);
                q++;
                while (q + 1 < n && cv[q].group > 1) q += 2;
                q++;  // proj
            }
        }
        if (starts.empty()) return "dp: no ConvFlow found";
        for (size_t j = 0; j < starts.size(); j++) {
            const std::string s = "dp/flows." + std::to_string(2 * (starts.size() - j) + 1);
            if (p != starts[j]) return "dp: ConvFlow walk lost its place";
            put(p++, s + "/pre");
            if (!dds(s + "/convs")) return s + ": no DDSConv";
            if (!is11(p)) return s + ": proj not found";
            put(p++, s + "/proj");
        }
    } else {  // plain DurationPredictor (models.py:120-165): [cond] conv_1 conv_2 proj
        if (gin) put(p++, "dp/cond");
        if (p + 2 >= n || cv[p].Cin != H || cv[p + 2].Cout != 1) return "dp: neither a stochastic nor a plain duration predictor";
        put(p, "dp/conv_1");
        put(p + 1, "dp/conv_2");
        put(p + 2, "dp/proj");
        p += 3;
    }
    const size_t dp_end = p;
    // ---- coupling flow (reverse order: flows 2(n-1), .., 2, 0): pre [cond_layer] (in_layers.i res_skip_layers.i)* post
    {
        std::vector<std::vector<size_t>> cps;
        if (C % 2) return "odd inter_channels";
        const int half = C / 2;
        while (p < n && is11(p) && cv[p].Cin == half) {
            std::vector<size_t> c;
            const int Hf = cv[p].Cout;
            c.push_back(p++);
            if (gin) c.push_back(p++);
            int i = 0;
            while (p + 1 < n && !cv[p].T && cv[p].Cin == Hf && cv[p].Cout == 2 * Hf && is11(p + 1) && cv[p + 1].Cin == Hf &&
                   !(cv[p].K == 1 && cv[p].Cout == half)) {
                c.push_back(p);
                c.push_back(p + 1);
                p += 2;
                i++;
            }
            if (!i || !is11(p) || cv[p].Cin != Hf || cv[p].Cout != half) return "flow: coupling layer without WN layers / post";
            c.push_back(p++);
            cps.push_back(c);
        }
        if (cps.empty()) return "flow: no coupling layer found";
        for (size_t j = 0; j < cps.size(); j++) {
            const std::string s = "flow/flows." + std::to_string(2 * (cps.size() - 1 - j));
            const auto &c = cps[j];
            size_t k = 0;
            put(c[k++], s + "/pre");
            if (gin) put(c[k++], s + "/enc/cond_layer");
            for (int i = 0; k + 1 < c.size(); i++, k += 2) {
                put(c[k], s + "/enc/in_layers." + std::to_string(i));
                put(c[k + 1], s + "/enc/res_skip_layers." + std::to_string(i));
            }
            put(c[k], s + "/post");
        }
    }
    // ---- generator: conv_pre [cond] (ups.s resblocks...)* conv_post
    if (p >= n || cv[p].T || cv[p].Cin != C) return "dec.conv_pre not found";
    put(p++, "dec/conv_pre");
    if (gin) {
        if (!is11(p)) return "dec.cond not found";
        put(p++, "dec/cond");
    }
    {
        int stage = 0, rb = 0;
        while (p < n && cv[p].T) {
            put(p++, "dec/ups." + std::to_string(stage++));
            std::vector<size_t> st;
            while (p < n && !cv[p].T && !(p == n - 1)) st.push_back(p++);
            if (st.empty()) return "dec: an upsampling stage without residual blocks";
            // split into residual blocks: a new block where the kernel size changes; equal kernels: by the period of the
            // dilation sequence
            std::vector<std::vector<size_t>> blocks;
            bool kchange = false;
            for (size_t i = 1; i < st.size(); i++) kchange = kchange || cv[st[i]].K != cv[st[0]].K;
            if (kchange) {
                for (size_t i = 0; i < st.size(); i++) {
                    if (i == 0 || cv[st[i]].K != cv[st[i - 1]].K) blocks.emplace_back();
                    blocks.back().push_back(st[i]);
                }
            } else {
                size_t per = st.size();
                for (size_t q = 1; q < st.size(); q++) {
                    if (st.size() % q) continue;
                    bool ok = true;
                    for (size_t i = q; i < st.size() && ok; i++) ok = cv[st[i]].dil == cv[st[i - q]].dil;
                    if (ok) {
                        per = q;
                        break;
                    }
                }
                for (size_t i = 0; i < st.size(); i++) {
                    if (i % per == 0) blocks.emplace_back();
                    blocks.back().push_back(st[i]);
                }
            }
            for (const auto &b : blocks) {
                bool t1 = b.size() % 2 == 0;  // ResBlock1: (convs1.j, convs2.j) pairs, every convs2 undilated (modules.py:220-298)
                for (size_t i = 1; i < b.size() && t1; i += 2) t1 = cv[b[i]].dil == 1;
                const std::string r = "dec/resblocks." + std::to_string(rb++);
                for (size_t i = 0; i < b.size(); i++)
                    put(b[i], t1 ? r + (i % 2 ? "/convs2." : "/convs1.") + std::to_string(i / 2) : r + "/convs." + std::to_string(i));
            }
        }
        if (!stage) return "dec: no upsampling stage";
    }
    if (p != n - 1 || cv[p].Cout != 1) return "dec.conv_post not found at the end of the graph";
    put(p++, "dec/conv_post");
    // ---- everything that hangs off a convolution's position: LayerNorm scale / shift, relative-position tables,
    // embeddings, the ElementwiseAffine of the duration flow
    std::string last_mod;
    int last_ch = 0, gathers = 0;
    bool got_g = false, got_b = false;
    const int dp_end_node = dp_end > 0 ? cv[dp_end - 1].node : -1;
    for (int i = 0; i < nn; i++) {
        const OnnxNode &nd = om.nodes[size_t(i)];
        if (!paths[size_t(i)].empty()) {
            std::string mod, leaf;
            split_path(paths[size_t(i)], mod, leaf);
            last_mod = mod;
            got_g = got_b = false;
            for (const auto &c : cv)
                if (c.node == i) last_ch = c.Cout;
            continue;
        }
        if (nd.op == "Gather" && !nd.inputs.empty()) {
            const OnnxTensor *t = om.find_init(nd.inputs[0]);
            if (t && t->dtype == 1 && t->dims.size() == 2) {
                if (gathers == 0) paths[size_t(i)] = "/enc_p/emb/Gather";
                else if (gathers == 1 && gin) paths[size_t(i)] = "/emb_g/Gather";
                gathers++;
            }
        } else if ((nd.op == "Mul" || nd.op == "Add") && !last_mod.empty()) {
            const OnnxTensor *t = nullptr;
            for (const auto &in : nd.inputs) {
                const OnnxTensor *c = om.find_init(in);
                if (c && c->dtype == 1 && c->dims.size() == 1 && c->dims[0] == last_ch && last_ch > 1) t = c;
            }
            if (!t) continue;
            std::string norm;
            auto tail = [&](const std::string &key, std::string &head, std::string &idx) {
                const size_t at = last_mod.rfind(key);
                if (at == std::string::npos) return false;
                head = last_mod.substr(0, at);
                idx = last_mod.substr(at + key.size());
                return true;
            };
            std::string head, idx;
            if (tail("attn_layers.", head, idx) && idx.size() > 7 && idx.compare(idx.size() - 7, 7, ".conv_o") == 0)
                norm = head + "norm_layers_1." + idx.substr(0, idx.size() - 7);
            else if (tail("ffn_layers.", head, idx) && idx.size() > 7 && idx.compare(idx.size() - 7, 7, ".conv_2") == 0)
                norm = head + "norm_layers_2." + idx.substr(0, idx.size() - 7);
            else if (tail("convs_sep.", head, idx)) norm = head + "norms_1." + idx;
            else if (tail("convs_1x1.", head, idx)) norm = head + "norms_2." + idx;
            else if (last_mod == "dp.conv_1") norm = "dp.norm_1";
            else if (last_mod == "dp.conv_2") norm = "dp.norm_2";
            if (norm.empty()) continue;
            if (nd.op == "Mul" && !got_g) got_g = true;
            else if (nd.op == "Add" && got_g && !got_b) got_b = true;
            else continue;
            for (auto &ch : norm)
                if (ch == '.') ch = '/';
            // (module paths use '/' between modules and '.' inside "name.index": restore the index dots)
            for (size_t k = 0; k + 1 < norm.size(); k++)
                if (norm[k] == '/' && std::isdigit((unsigned char)norm[k + 1])) norm[k] = '.';
            paths[size_t(i)] = "/" + norm + "/" + nd.op;
        } else if (nd.op == "Pad" && !nd.inputs.empty() && last_mod.find("attn_layers.") != std::string::npos) {
            const OnnxTensor *t = om.find_init(nd.inputs[0]);
            if (t && t->dtype == 1 && t->dims.size() == 3) {
                std::string a = last_mod.substr(0, last_mod.rfind('.'));  // enc_p.encoder.attn_layers.N
                for (auto &ch : a)
                    if (ch == '.') ch = '/';
                for (size_t k = 0; k + 1 < a.size(); k++)
                    if (a[k] == '/' && std::isdigit((unsigned char)a[k + 1])) a[k] = '.';
                paths[size_t(i)] = "/" + a + "/Pad";
            }
        } else if ((nd.op == "Sub" || nd.op == "Exp") && i > dp_end_node && dp_end_node >= 0 && last_mod.rfind("dp.", 0) == 0) {
            for (const auto &in : nd.inputs) {
                const OnnxTensor *t = om.find_init(in);
                if (t && t->dtype == 1 && t->numel() == 2) paths[size_t(i)] = "/dp/flows.0/" + nd.op;
            }
        }
    }
    return "";
}

void resolve(const OnnxModel &om, Resolver &R) {
    auto init = [&](const std::string &n) -> const OnnxTensor * { return om.find_init(n); };
    std::map<std::string, int> pad_seen;
    // names: the exporter's module paths where the graph has them, else the structural walk's; both present: they agree
    bool named = false;
    for (const auto &n : om.nodes)
        named = named || ((n.op == "Conv" || n.op == "ConvTranspose") && n.name.size() > 1 && n.name[0] == '/' &&
                          n.name.find('/', 1) != std::string::npos);
    std::vector<std::string> sp;
    const std::string serr = structural_paths(om, sp);
    static const bool force_struct = std::getenv("VITSMI_RESOLVE_BY_STRUCTURE") != nullptr;  // tests
    const bool use_struct = !named || force_struct;
    if (use_struct && !serr.empty())
        throw std::runtime_error("the graph's nodes carry no module-path names and its structure is not that of a VITS export: " + serr);
    if (named && serr.empty()) {
        // The structural walk is a heuristic (block boundaries from kernel / dilation changes, emb_g = the second 2-D Gather):
        // where a graph carries the exporter's module paths those are the authority, and a disagreement is reported once, not
        // fatal - a well-named third-party export must keep loading.  VITSMI_STRICT_NAMES=1 turns it into an error (tests,
        // debugging a suspicious file).
        static const bool strict = std::getenv("VITSMI_STRICT_NAMES") != nullptr;
        for (size_t i = 0; i < om.nodes.size(); i++) {
            const OnnxNode &n = om.nodes[i];
            if ((n.op != "Conv" && n.op != "ConvTranspose") || sp[i].empty()) continue;
            if (sp[i] != n.name) {
                const std::string msg = "node '" + n.name + "' sits where the graph's structure expects '" + sp[i] + "'";
                if (strict) throw std::runtime_error(msg);
                // every disagreement is kept for the caller (vits_meta "vitsmi.name_warnings", one per line); stderr names the
                // first and says how many there are
                if (R.name_warnings.empty())
                    fprintf(stderr, "vitsmi: warning: %s (names win; VITSMI_STRICT_NAMES=1 rejects such files; the full list: "
                                    "vits_meta \"vitsmi.name_warnings\")\n", msg.c_str());
                R.name_warnings += msg + "\n";
            }
        }
    }
    for (size_t ni = 0; ni < om.nodes.size(); ni++) {
        const auto &n = om.nodes[ni];
        const std::string &nm = use_struct ? sp[ni] : n.name;
        if (nm.empty()) continue;
        std::string mod, leaf;
        split_path(nm, mod, leaf);
        if (n.op == "Conv" || n.op == "ConvTranspose") {
            if (n.inputs.size() > 1 && !init(n.inputs[1]))
                // (the reference exports with constant folding, export_onnx.py:318-327, which turns the weight-norm chains
                // g * v / |v| of the flow's convs into plain initializers; this reader does not evaluate graph arithmetic)
                throw std::runtime_error("the weight of node '" + n.name + "' is computed inside the graph (weight-norm chain left in: "
                                         "exported with do_constant_folding=False?); re-export with constant folding");
            if (n.inputs.size() > 1) R.put(mod + ".weight", init(n.inputs[1]));
            if (n.inputs.size() > 2 && !n.inputs[2].empty()) R.put(mod + ".bias", init(n.inputs[2]));
            auto a = n.ints.find("dilations");
            if (a != n.ints.end() && !a->second.empty()) R.ints[mod + ".dilation"] = a->second[0];
            a = n.ints.find("strides");
            if (a != n.ints.end() && !a->second.empty()) R.ints[mod + ".stride"] = a->second[0];
            a = n.ints.find("pads");
            if (a != n.ints.end() && !a->second.empty()) R.ints[mod + ".pad"] = a->second[0];
            a = n.ints.find("group");
            R.ints[mod + ".group"] = (a != n.ints.end() && !a->second.empty()) ? a->second[0] : 1;
        } else if (n.op == "Gather" && !n.inputs.empty()) {
            const OnnxTensor *t = init(n.inputs[0]);
            if (t && t->dtype == 1 && t->dims.size() == 2) R.put(mod + ".weight", t);
        } else if (n.op == "LayerNormalization" && n.inputs.size() >= 3) {
            // opset >= 17: modules.py:14-26 (F.layer_norm over the channel axis) is exported as ONE node (X, Scale, B) instead of
            // the ReduceMean / Sub / Pow / .. / Mul / Add chain of opset <= 16
            const OnnxTensor *g = init(n.inputs[1]), *bt = init(n.inputs[2]);
            if (g && g->dtype == 1 && g->dims.size() == 1) R.put(mod + ".gamma", g);
            if (bt && bt->dtype == 1 && bt->dims.size() == 1) R.put(mod + ".beta", bt);
            auto ep = n.floats.find("epsilon");
            if (ep != n.floats.end() && std::fabs(ep->second - 1e-5f) > 1e-7f)
                throw std::runtime_error("LayerNormalization node '" + n.name + "' has epsilon " + std::to_string(ep->second) +
                                         "; this engine implements the reference's 1e-5 (modules.py:17)");
        } else if ((n.op == "Mul" || n.op == "Add") && mod.find("norm") != std::string::npos) {
            for (const auto &i : n.inputs) {
                const OnnxTensor *t = init(i);
                if (t && t->dtype == 1 && t->dims.size() == 1) R.put(mod + (n.op == "Mul" ? ".gamma" : ".beta"), t);
            }
        } else if (n.op == "Pad" && mod.find("attn_layers") != std::string::npos && !n.inputs.empty()) {
            const OnnxTensor *t = init(n.inputs[0]);
            if (t && t->dtype == 1 && t->dims.size() == 3) {
                int k = pad_seen[mod]++;
                R.put(mod + (k == 0 ? ".emb_rel_k" : ".emb_rel_v"), t);
            }
        } else if (mod == "dp.flows.0" && n.op == "Sub") {
            for (const auto &i : n.inputs) R.put("dp.flows.0.m", init(i));
        } else if (mod == "dp.flows.0" && n.op == "Exp") {
            for (const auto &i : n.inputs) {
                const OnnxTensor *t = init(i);
                if (t && t->data() && !R.t.count("dp.flows.0.logs")) {
                    // the exporter constant-folded Neg(logs): undo it
                    R.owned.emplace_back(t->data(), t->data() + t->numel());
                    for (auto &v : R.owned.back()) v = -v;
                    TRef r;
                    r.p = R.owned.back().data();
                    r.dims = t->dims;
                    R.t["dp.flows.0.logs"] = r;
                }
            }
        }
    }
    // relative-position tables are also reachable by parameter name (they are never folded)
    for (const auto &kv : om.init) {
        const std::string &k = kv.first;
        auto ends = [&](const char *s) {
            size_t n = std::strlen(s);
            return k.size() >= n && k.compare(k.size() - n, n, s) == 0;
        };
        if (ends("emb_rel_k") || ends("emb_rel_v")) R.put(k, &kv.second);
    }
}

// ---------------------------------------------------------------------------------- packing

// `dry`: lay the arena out (offsets, descriptors, total size) without materialising a single weight: what a handle
// needs when the packed bytes already sit on its GPU (vits_open_with_arena: RCCL-broadcast weights, or a second
// handle sharing the first one's arena).
struct Packer {
    std::vector<float> &arena;
    const bool dry;
    int64_t size = 0;  // floats laid out so far (== arena.size() unless dry)
    explicit Packer(std::vector<float> &a, bool dry_ = false) : arena(a), dry(dry_), size(int64_t(a.size())) {}
    int64_t alloc(int64_t n) {
        int64_t off = (size + 63) / 64 * 64;  // 256-byte alignment
        size = off + n;
        if (!dry) arena.resize(size_t(size), 0.f);
        return off;
    }
    int64_t put(const float *p, int64_t n) {
        int64_t off = alloc(n);
        if (!dry) std::memcpy(arena.data() + off, p, size_t(n) * 4);
        return off;
    }
    int64_t put(const TRef &r) { return put(r.p, r.numel()); }
};

// ---- tile / chunk selection (mirrors conv_engine.hip.hpp; kept here so model.cpp stays HIP-free)
// cfg: 0: 32x512, 1: 64x256, 2: 128x128 (long sequences), 3: 64x64, 4: 32x128 (short sequences), 5: 32x64 with the
// reduction split over the workgroup's four waves (token domain)
int tile_m(int cfg) { return cfg == 2 ? 128 : ((cfg == 1 || cfg == 3) ? 64 : 32); }
int tile_n(int cfg) {
    static const int n[6] = {512, 256, 128, 64, 128, 64};
    return n[cfg];
}
// floats of ONE pipeline stage (x tile + A slab) in the 16-byte-DMA layout (the larger one)
size_t stage_floats(int cfg, int K, int dil, int padL, int CK) {
    const int BN = tile_n(cfg), BM = tile_m(cfg), halo = (K - 1) * dil;
    const int padLa = (padL + 3) & ~3, padRa = (halo - padL + 3) & ~3;
    const size_t LW = size_t(BN + padLa + padRa);
    const size_t xs = (size_t(CK) * LW + 1023) / 1024 * 1024;
    const size_t as = size_t(BM / 32) * size_t(K * CK / 8) * 256;
    return xs + as;
}
size_t stage_capacity(int cfg) {  // conv_engine.hip.hpp conv_stage_floats
    static const long cap3 = [] {
        const char *e = std::getenv("VITSMI_STAGE_CAP_SMALL");  // tuning experiments only
        return e ? std::atol(e) : 4864l;
    }();
    if (cfg == 5) return 6144;  // (its workgroups hold >= 32 KiB for the partial tiles anyway: room for 32-channel chunks)
    return cfg <= 2 ? 9728 : size_t(cap3);
}

// hint: 0 = frame/sample domain (generator), 1 = frame domain (flow), 2 = token domain (encoder, durations)
thread_local int t_cfg_override = -1, t_ck_override = -1;  // kernel tuning only (tools/conv_bench.py)

void pick_tiling(int Cin, int Cout, int K, int dil, int padL, int hint, int &cfg, int &CK) {
    if (t_cfg_override >= 0) {
        cfg = t_cfg_override;
        CK = t_ck_override > 0 ? t_ck_override : 8;
        if (stage_floats(cfg, K, dil, padL, CK) > stage_capacity(cfg)) throw std::runtime_error("override does not fit LDS");
        return;
    }
    std::vector<int> cands;
    static const bool no_splitk = std::getenv("VITSMI_NO_SPLITK") != nullptr;  // A/B timing only
    // token domain: deep reductions into few rows (the encoder's 768 -> 192, k = 3 FFN conv: 2304 products per output)
    // split the reduction over the workgroup's waves (cfg 5); wide, shallow layers keep a block per wave (measured at
    // batch 32: 127 -> 98 us for the former, 94 -> 104 us for the 192 -> 768 conv)
    if (hint == 2) cands = {Cout <= 32 ? 4 : ((!no_splitk && Cin * K >= 1024) ? 5 : 3)};
    else if (hint == 1 && Cout % 128 != 0) cands = {Cout <= 32 ? 4 : 3};
    // long-sequence tiles, also the fallback when a wide kernel does not fit a small tile's LDS stage
    if (Cout <= 32) cands.push_back(0);
    else if (Cout % 128 == 0) {
        cands.push_back(2);
        cands.push_back(1);
    } else
        cands.push_back(1);
    const int cin8 = (Cin + 7) / 8 * 8;
    // first choice: a stage of <= 26 KiB, so that three workgroups (12 waves) share a CU; the widest kernels
    // (k = 11) only fit the 38 KiB stage (two workgroups)
    for (size_t cap : {size_t(6656), size_t(0)})
        for (int c : cands)
            for (int ck : {32, 16, 8}) {
                if (ck > cin8 && ck != 8) continue;
                const size_t lim = (cap && c <= 2) ? cap : stage_capacity(c);
                if (stage_floats(c, K, dil, padL, ck) <= lim) {
                    cfg = c;
                    CK = ck;
                    return;
                }
            }
    throw std::runtime_error("conv receptive field too wide for the LDS stage (kernel " + std::to_string(K) +
                             ", dilation " + std::to_string(dil) + ")");
}

thread_local int t_hint = 0;  // size class of the layers being packed (set by Model::build)
thread_local bool t_sx_f16 = false;  // pack_conv_sx: two scaled fp16 planes instead of three bf16 planes
thread_local bool t_sx_force16 = false;  // pack_conv_sx: the 16x16x32 layout whatever the channel count (conv_sx_pair16's weights)
thread_local bool t_sx_h1 = false;   // pack_conv_sx: ONE scaled fp16 plane in the 16x16x32 layout (VITSMI_GEN_PRECISION=f16)
// the encoder / flow run f16x3 under the default arithmetic AND under the reduced-precision vocoder ("f16": everything in
// front of z is unchanged); bf16x6 keeps them on their exact engines
bool precision_is_fp16_family(const char *pe) { return !pe || !*pe || std::string(pe) == "f16x3" || std::string(pe) == "f16"; }
thread_local int t_sx_min_cfg = 0;   // pack_conv_sx: smallest tile index allowed (1 = no 128-row tiles)
thread_local bool t_sx_shape32 = false;  // pack_conv_sx: never the 16x16x32 packing (bench / ablation hooks)

// W is addressed through a functor so that permutations / transposed-conv rewrites need no copies:
// w(co, ci, tap) for co < Cout, ci < Cin, tap < K.
template <class WF>
ConvDesc pack_conv(Packer &P, int Cin, int Cout, int K, int dil, int padL, WF w, const float *bias_virtual) {
    const int hint = t_hint;
    ConvDesc d;
    d.Cin = Cin;
    d.Cout = Cout;
    d.K = K;
    d.dil = dil;
    d.padL = padL;
    pick_tiling(Cin, Cout, K, dil, padL, hint, d.cfg, d.CK);
    d.nchunks = (Cin + d.CK - 1) / d.CK;
    int bm = tile_m(d.cfg);
    d.mblocks = (Cout + bm - 1) / bm * (bm / 32);
    int64_t per_block = int64_t(d.nchunks * K * d.CK / 8) * 64 * 4;  // float4 groups x lanes x 4
    d.w_off = P.alloc(per_block * d.mblocks);
    if (bias_virtual) d.b_off = P.put(bias_virtual, Cout);
    d.macs_per_t = double(Cin) * Cout * K;
    if (P.dry) return d;
    float *dst = P.arena.data() + d.w_off;
    int half = d.CK / 2;
    // Layout: Wp[m-tile][chunk][block-in-tile][group][lane][4]: the A slab one workgroup needs for one
    // chunk (MB blocks x spc groups x 1 KiB) is one contiguous range -> LDS-DMA pieces are base + i KiB.
    const int MB = bm / 32, spc = K * d.CK / 8;
    for (int mb = 0; mb < d.mblocks; mb++)
        for (int chunk = 0; chunk < d.nchunks; chunk++)
            for (int tap = 0; tap < K; tap++)
                for (int pair = 0; pair < half; pair++) {
                    int step = tap * half + pair;  // k-step inside the chunk
                    int64_t slab = ((int64_t(mb / MB) * d.nchunks + chunk) * MB + (mb % MB)) * spc;
                    float *g = dst + (slab + step / 4) * 256 + (step & 3);
                    for (int lane = 0; lane < 64; lane++) {
                        int co = mb * 32 + (lane & 31);
                        int ci = chunk * d.CK + 2 * pair + (lane >> 5);
                        g[lane * 4] = (co < Cout && ci < Cin) ? w(co, ci, tap) : 0.f;
                    }
                }
    return d;
}

// plain Conv1d module "<name>": weight [Cout, Cin, K]
ConvDesc pack_named(Packer &P, const Resolver &R, const std::string &name, int dil, int padL,
                    const std::vector<int> *in_perm = nullptr, const std::vector<int> *out_perm = nullptr) {
    const TRef &w = R.need(name + ".weight", 3);
    int Cout = int(w.dims[0]), Cin = int(w.dims[1]), K = int(w.dims[2]);
    if (R.geti(name + ".group", 1) != 1) throw std::runtime_error(name + ": grouped conv not expected here");
    const TRef *b = R.bias_of(name + ".bias", Cout);
    if ((in_perm && int(in_perm->size()) != Cin) || (out_perm && int(out_perm->size()) != Cout))
        throw std::runtime_error(name + ": channel count does not match the coupling layer's half width");
    const float *wp = w.p;
    auto wf = [&](int co, int ci, int tap) {
        int so = out_perm ? (*out_perm)[co] : co;
        int si = in_perm ? (*in_perm)[ci] : ci;
        return wp[(int64_t(so) * Cin + si) * K + tap];
    };
    std::vector<float> bperm;
    const float *bp = b ? b->p : nullptr;
    if (b && out_perm) {
        bperm.resize(Cout);
        for (int c = 0; c < Cout; c++) bperm[c] = b->p[(*out_perm)[c]];
        bp = bperm.data();
    }
    return pack_conv(P, Cin, Cout, K, dil, padL, wf, bp);
}

// ConvTranspose1d [Cin, Cout, K], stride u, padding p  ->  dense conv with Cout*u virtual
// channels (co' = co*u + r) over taps o_min..o_max and a pixel-shuffle store.
ConvDesc pack_convT(Packer &P, const Resolver &R, const std::string &name) {
    const TRef &w = R.need(name + ".weight", 3);
    int Cin = int(w.dims[0]), Cout = int(w.dims[1]), K = int(w.dims[2]);
    int u = int(R.geti(name + ".stride", -1));
    int p = int(R.geti(name + ".pad", -1));
    if (u < 1 || p < 0) throw std::runtime_error(name + ": missing stride/pads attributes");
    if (K - 2 * p != u) throw std::runtime_error(name + ": transposed conv with K-2*pad != stride is unsupported");
    int o_max = -1000000, o_min = 1000000;
    for (int r = 0; r < u; r++) {
        int j0 = (r + p) % u, e = (r + p) / u;
        int M = (K - j0 + u - 1) / u;
        if (M <= 0) continue;
        if (e > o_max) o_max = e;
        if (e - (M - 1) < o_min) o_min = e - (M - 1);
    }
    int Kv = o_max - o_min + 1;
    const float *wp = w.p;
    auto wf = [&](int cov, int ci, int tapv) -> float {
        int co = cov / u, r = cov % u;
        int j0 = (r + p) % u, e = (r + p) / u;
        int m = e - o_min - tapv;
        int j = j0 + m * u;
        if (m < 0 || j >= K) return 0.f;
        return wp[(int64_t(ci) * Cout + co) * K + j];
    };
    std::vector<float> bv;
    const TRef *b = R.bias_of(name + ".bias", Cout);
    if (b) {
        bv.resize(size_t(Cout) * u);
        for (int c = 0; c < Cout * u; c++) bv[c] = b->p[c / u];
    }
    ConvDesc d = pack_conv(P, Cin, Cout * u, Kv, 1, -o_min, wf, b ? bv.data() : nullptr);
    d.ups = u;
    d.macs_per_t = double(Cin) * Cout * K;  // per INPUT time step
    return d;
}

// ---- split-operand (sx) packing: weights as three bf16 planes (or two scaled fp16 planes) in the A-operand lane order of
// v_mfma_f32_32x32x16_bf16 (lane l: row l&31, k = 8*(l>>5) .. +7 = eight consecutive input channels).
int sx_tile_m(int cfg) { return cfg == 0 ? 128 : (cfg == 1 ? 64 : 32); }
int sx_tile_n(int) { return 256; }
int sx_pick_cfg(int Cout) {
    static const int min_cfg = [] {
        const char *e = std::getenv("VITSMI_SX_MIN_CFG");  // tuning experiments only: 1 = no 128-row tiles
        return e ? std::atoi(e) : 0;
    }();
    static const int cfg1_for = [] {
        const char *e = std::getenv("VITSMI_SX_CFG1_FOR");  // tuning experiments only: this Cout gets 64-row tiles
        return e ? std::atoi(e) : -1;
    }();
    int cfg = Cout % 128 == 0 ? 0 : (Cout % 64 == 0 ? 1 : 2);
    if (Cout == cfg1_for && cfg == 0) cfg = 1;
    const int lo = min_cfg > t_sx_min_cfg ? min_cfg : t_sx_min_cfg;
    return cfg < lo ? lo : cfg;
}

template <class WF>
ConvDesc pack_conv_sx(Packer &P, int Cin, int Cout, int K, int dil, int padL, WF w, const float *bias_virtual) {
    if (!sx_supported(Cin, Cout, Cout, K, dil)) throw std::runtime_error("conv shape not supported by the sx engine");
    // the 32x32x16 pipeline needs >= 3 taps per chunk: narrower kernels get zero taps appended on the right.  (The
    // 16x16x32 loop waits for everything at each chunk start and takes a 1 x 1 conv as it is: decided below.)
    const int Kreal = K;
    static const bool shape32_only = [] {
        const char *e = std::getenv("VITSMI_SX_SHAPE");  // "32": A/B timing against the v_mfma_f32_32x32x16 loop
        return e && std::string(e) == "32";
    }();
    const bool h1 = t_sx_h1;
    if (h1 && (Cin % 32 || (size_t)4 * (256 + (Kreal - 1) * dil) * 16 > (size_t)10 * 4096))
        throw std::runtime_error("the f16 single-plane arithmetic needs Cin % 32 == 0 and a halo of at most 384 columns");
    // (force16: the 32- / 64-channel convs of a plane-stream generator - the fused pair kernel's weights; their own x stage
    // of up to twelve DMA rounds serves the unfused fallback)
    const bool want16 = h1 || (t_sx_f16 && t_sx_force16 && sx_raw_format(Cin) && Cin % 32 == 0 &&
                               (size_t)8 * (256 + (Kreal - 1) * dil) * 16 <= (size_t)12 * 4096) ||
                        (t_sx_f16 && !sx_raw_format(Cin) && !shape32_only && !t_sx_shape32 && Cin % 32 == 0 &&
                         (size_t)8 * (256 + (Kreal - 1) * dil) * 16 <= (size_t)10 * 4096);
    if (K < 3 && !want16) K = 3;
    auto wz = [&](int co, int ci, int tap) { return tap < Kreal ? w(co, ci, tap) : 0.f; };
    ConvDesc d;
    d.sx = true;
    d.Cin = Cin;
    d.Cout = Cout;
    d.K = K;
    d.dil = dil;
    d.padL = padL;
    d.CK = 16;
    d.nchunks = Cin / 16;
    d.rawin = sx_raw_format(Cin) && !h1 && !(t_sx_f16 && t_sx_force16);  // (plane-stream generators: every tensor is a plane tensor)
    d.h1 = h1;
    d.cfg = sx_pick_cfg(Cout);
    if (d.rawin && d.cfg == 0) d.cfg = 1;  // the raw-input path exists for the 64- and 32-row tiles only
    d.mblocks = Cout / 32;
    const int MB = sx_tile_m(d.cfg) / 32;
    const int npw = h1 ? 1 : (t_sx_f16 ? 2 : 3);                   // planes per 32-row block
    // 16x16x32 main loop (f16x3, plane input, 32-channel chunks): when the x stage of a 32-channel chunk (8 rows of
    // 256 + halo cells) fits ten DMA rounds, i.e. two workgroups per CU
    d.s16 = want16;
    if (d.s16) {
        d.CK = 32;
        d.nchunks = Cin / 32;
    }
    const int64_t kib = int64_t(d.mblocks) * (Cin / 16) * K * npw;  // 1 KiB = one (block, plane) fragment set of k = 16
    d.w_off = P.alloc(kib * 256);
    if (bias_virtual) d.b_off = P.put(bias_virtual, Cout);
    d.macs_per_t = double(Cin) * Cout * Kreal;
    float wmul = 1.f;
    if (t_sx_f16 || h1) {
        // per-tensor power of two that lifts the largest weight into [2^14, 2^15): both fp16 planes of every weight
        // within 2^-18 of the largest are then normal numbers; undone exactly on the accumulators
        float wmax = 0.f;
        for (int co = 0; co < Cout; co++)
            for (int ci = 0; ci < Cin; ci++)
                for (int tap = 0; tap < Kreal; tap++) {
                    const float a = std::fabs(w(co, ci, tap));
                    if (std::isfinite(a) && a > wmax) wmax = a;
                }
        int e = 0;
        if (wmax > 0.f) std::frexp(wmax, &e);  // wmax = m * 2^e, m in [0.5, 1)
        wmul = std::ldexp(1.f, 15 - e);
        d.f16 = !h1;
        d.wscale = std::ldexp(1.f, e - 15);
    }
    if (P.dry) return d;
    uint16_t *dst = reinterpret_cast<uint16_t *>(P.arena.data() + d.w_off);
    if (d.s16) {
        // [m-tile][chunk of 32 ci][tap][32-row block][sub-block a][plane][lane][8]: 4 KiB per (32-row block, step)
        // (single-plane mode: one plane, 2 KiB)
        for (int mb = 0; mb < d.mblocks; mb++)
            for (int chunk = 0; chunk < d.nchunks; chunk++)
                for (int tap = 0; tap < K; tap++) {
                    const int64_t base = ((((int64_t)(mb / MB) * d.nchunks + chunk) * K + tap) * MB + (mb % MB)) * 2 * npw;
                    for (int sub = 0; sub < 2; sub++)
                        for (int lane = 0; lane < 64; lane++) {
                            const int i16 = lane & 15, q = i16 >> 2;
                            const int row32 = (i16 & 3) + 8 * (2 * sub + (q & 1)) + 4 * (q >> 1);
                            for (int i = 0; i < 8; i++) {
                                uint16_t p[3];
                                split2h_host(wz(mb * 32 + row32, chunk * 32 + 8 * (lane >> 4) + i, tap) * wmul, p);
                                for (int pl = 0; pl < npw; pl++) dst[(base + sub * npw + pl) * 512 + lane * 8 + i] = p[pl];
                            }
                        }
                }
        return d;
    }
    for (int mb = 0; mb < d.mblocks; mb++)
        for (int chunk = 0; chunk < d.nchunks; chunk++)
            for (int tap = 0; tap < K; tap++) {
                const int64_t base = ((((int64_t)(mb / MB) * d.nchunks + chunk) * K + tap) * MB + (mb % MB)) * npw;
                for (int lane = 0; lane < 64; lane++)
                    for (int i = 0; i < 8; i++) {
                        uint16_t p[3];
                        const float wv = wz(mb * 32 + (lane & 31), chunk * 16 + 8 * (lane >> 5) + i, tap);
                        if (t_sx_f16) split2h_host(wv * wmul, p);
                        else split3_host(wv, p);
                        for (int pl = 0; pl < npw; pl++) dst[(base + pl) * 512 + lane * 8 + i] = p[pl];
                    }
            }
    return d;
}

// out_perm (optional): packed row r holds the module's output channel out_perm[r]
ConvDesc pack_named_sx(Packer &P, const Resolver &R, const std::string &name, int dil, int padL,
                       const std::vector<int> *out_perm = nullptr) {
    const TRef &w = R.need(name + ".weight", 3);
    int Cout = int(w.dims[0]), Cin = int(w.dims[1]), K = int(w.dims[2]);
    if (R.geti(name + ".group", 1) != 1) throw std::runtime_error(name + ": grouped conv not expected here");
    const TRef *b = R.bias_of(name + ".bias", Cout);
    if (out_perm && int(out_perm->size()) != Cout) throw std::runtime_error(name + ": bad row permutation");
    const float *wp = w.p;
    auto wf = [&](int co, int ci, int tap) { return wp[(int64_t(out_perm ? (*out_perm)[co] : co) * Cin + ci) * K + tap]; };
    std::vector<float> bperm;
    const float *bp = b ? b->p : nullptr;
    if (b && out_perm) {
        bperm.resize(Cout);
        for (int c = 0; c < Cout; c++) bperm[c] = b->p[(*out_perm)[c]];
        bp = bperm.data();
    }
    return pack_conv_sx(P, Cin, Cout, K, dil, padL, wf, bp);
}

// geometry of ConvTranspose1d [Cin, Cout, K] (stride u, padding p) as a dense conv over taps o_min..o_max
struct ConvTGeom {
    int Cin, Cout, K, u, p, o_min, o_max;
};
ConvTGeom convt_geom(const Resolver &R, const std::string &name) {
    const TRef &w = R.need(name + ".weight", 3);
    ConvTGeom g;
    g.Cin = int(w.dims[0]);
    g.Cout = int(w.dims[1]);
    g.K = int(w.dims[2]);
    g.u = int(R.geti(name + ".stride", -1));
    g.p = int(R.geti(name + ".pad", -1));
    if (g.u < 1 || g.p < 0) throw std::runtime_error(name + ": missing stride/pads attributes");
    if (g.K - 2 * g.p != g.u) throw std::runtime_error(name + ": transposed conv with K-2*pad != stride is unsupported");
    g.o_max = -1000000;
    g.o_min = 1000000;
    for (int r = 0; r < g.u; r++) {
        int j0 = (r + g.p) % g.u, e = (r + g.p) / g.u;
        int M = (g.K - j0 + g.u - 1) / g.u;
        if (M <= 0) continue;
        if (e > g.o_max) g.o_max = e;
        if (e - (M - 1) < g.o_min) g.o_min = e - (M - 1);
    }
    return g;
}

// Same rewrite as pack_convT, but virtual rows are ordered (block of 32 channels, phase r, channel): a 32-row block
// holds 32 consecutive real channels of ONE output phase r, which is what the sx epilogue's cell stores need, and the
// blocks of a tile are the phases of the same channels, so that a workgroup writes whole runs of an output line
// (conv_sx_engine.hip.hpp, geom).
ConvDesc pack_convT_sx(Packer &P, const Resolver &R, const std::string &name) {
    const ConvTGeom g = convt_geom(R, name);
    const float *wp = R.req(name + ".weight").p;
    const int Kv = g.o_max - g.o_min + 1;
    auto row_of = [&](int cov, int &r, int &co) {
        const int cb = cov / (32 * g.u), rem = cov % (32 * g.u);
        r = rem / 32;
        co = cb * 32 + rem % 32;
    };
    auto wf = [&](int cov, int ci, int tapv) -> float {
        int r, co;
        row_of(cov, r, co);
        int j0 = (r + g.p) % g.u, e = (r + g.p) / g.u;
        int m = e - g.o_min - tapv;
        int j = j0 + m * g.u;
        if (m < 0 || j >= g.K) return 0.f;
        return wp[(int64_t(ci) * g.Cout + co) * g.K + j];
    };
    std::vector<float> bv;
    const TRef *b = R.bias_of(name + ".bias", g.Cout);
    if (b) {
        bv.resize(size_t(g.Cout) * g.u);
        for (int c = 0; c < g.Cout * g.u; c++) {
            int r, co;
            row_of(c, r, co);
            bv[c] = b->p[co];
        }
    }
    if (g.Cout % 32) throw std::runtime_error(name + ": sx transposed conv needs Cout % 32 == 0");
    ConvDesc d = pack_conv_sx(P, g.Cin, g.Cout * g.u, Kv, 1, -g.o_min, wf, b ? bv.data() : nullptr);
    d.ups = g.u;
    d.macs_per_t = double(g.Cin) * g.Cout * g.K;
    // K = 2 u, three dense taps starting at -1: phase r reads taps {e, e + 1} with e = (r + p) / u in {0, 1} - the third is zero
    if (g.K == 2 * g.u && Kv == 3 && g.o_min == -1 && d.K == 3) {
        bool ok = true;  // (checked against the packed function itself, tap by tap)
        for (int cov = 0; cov < g.Cout * g.u && ok; cov += 32) {
            const int r = (cov / 32) % g.u, zt = (r + g.p) / g.u == 0 ? 2 : 0;
            for (int ci = 0; ci < g.Cin && ok; ci++)
                for (int k = 0; k < 32 && ok; k++) ok = wf(cov + k, ci, zt) == 0.f;
        }
        if (ok) d.zt_p = g.p;
    }
    return d;
}

int same_pad(int K, int dil) { return (K * dil - dil) / 2; }  // commons.py:17-18, modules.py:102,163

DDSDesc pack_dds(Packer &P, const Resolver &R, const std::string &pfx) {
    DDSDesc d;
    for (int l = 0; l < 4; l++) {
        std::string s = pfx + ".convs_sep." + std::to_string(l);
        if (!R.get(s + ".weight")) break;
        const TRef *w = &R.need(s + ".weight", 3);  // depthwise [C, 1, K]
        if (w->dims[1] != 1) throw std::runtime_error(s + ": depthwise conv expected");
        const int64_t Cd = w->dims[0];
        d.n_layers = l + 1;
        d.K = int(w->dims[2]);
        auto &L = d.l[l];
        L.dw_w = P.put(*w);
        L.dw_b = P.put(R.need(s + ".bias", 1, Cd));
        int dil = 1;
        for (int i = 0; i < l; i++) dil *= d.K;  // modules.py:101
        L.dil = int(R.geti(s + ".dilation", dil));
        L.ln1_g = P.put(R.need(pfx + ".norms_1." + std::to_string(l) + ".gamma", 1, Cd));
        L.ln1_b = P.put(R.need(pfx + ".norms_1." + std::to_string(l) + ".beta", 1, Cd));
        L.ln2_g = P.put(R.need(pfx + ".norms_2." + std::to_string(l) + ".gamma", 1, Cd));
        L.ln2_b = P.put(R.need(pfx + ".norms_2." + std::to_string(l) + ".beta", 1, Cd));
        L.pw = pack_named(P, R, pfx + ".convs_1x1." + std::to_string(l), 1, 0);
        if (L.pw.Cin != Cd || L.pw.Cout != Cd) throw std::runtime_error(pfx + ": 1x1 conv does not match the depthwise width");
        if (Cd % 16 == 0) {
            // the 16-column kernel's A fragments: lane (row & 15, k & 3) reads its row's weights of k = 4 step + (k & 3), four
            // consecutive steps per float4: Wp[((row tile * 4 + k & 3) * 16 + row & 15) * (C / 4) + step]
            const TRef &w1 = R.need(pfx + ".convs_1x1." + std::to_string(l) + ".weight", 3);
            const int C = int(Cd), NS = C / 4;
            L.pw16 = P.alloc(int64_t(C) * C);
            if (!P.dry) {
                float *dst = P.arena.data() + L.pw16;
                for (int o = 0; o < C; o++)
                    for (int i = 0; i < C; i++)
                        dst[(int64_t((o / 16) * 4 + (i & 3)) * 16 + (o & 15)) * NS + (i >> 2)] = w1.p[int64_t(o) * C + i];
            }
        }
    }
    if (!d.n_layers) throw std::runtime_error("no DDSConv layers under " + pfx);
    return d;
}

// A 1 x 1 conv [Cout, Cin, 1] in the A-operand layout of dds_layer16_kernel (see pack_dds: pw16), rows zero-padded to a
// multiple of 16: the conv that follows a DDSConv stack runs as the tail of its last layer.  -1 where the shape does not fit.
int64_t pack_pw16(Packer &P, const Resolver &R, const std::string &name, int Cin) {
    const TRef *w = R.get(name + ".weight");
    if (!w || w->dims.size() != 3 || w->dims[2] != 1 || w->dims[1] != Cin || Cin % 16 || R.geti(name + ".group", 1) != 1) return -1;
    const int Cout = int(w->dims[0]), Cp = (Cout + 15) / 16 * 16, NS = Cin / 4;
    const int64_t off = P.alloc(int64_t(Cp) * Cin);
    if (!P.dry) {
        float *dst = P.arena.data() + off;
        for (int64_t i = 0; i < int64_t(Cp) * Cin; i++) dst[i] = 0.f;
        for (int o = 0; o < Cout; o++)
            for (int i = 0; i < Cin; i++)
                dst[(int64_t((o / 16) * 4 + (i & 3)) * 16 + (o & 15)) * NS + (i >> 2)] = w->p[int64_t(o) * Cin + i];
    }
    return off;
}

}  // namespace

uint16_t bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return uint16_t((u >> 16) | 0x40);  // NaN stays NaN
    return uint16_t((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

float bf16_to_f32(uint16_t h) {
    uint32_t u = uint32_t(h) << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

void split3_host(float v, uint16_t p[3]) {
    p[0] = bf16_rne(v);
    const float r1 = v - bf16_to_f32(p[0]);
    p[1] = bf16_rne(r1);
    const float r2 = r1 - bf16_to_f32(p[1]);
    p[2] = bf16_rne(r2);
}

uint16_t f16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    const uint16_t sign = uint16_t((u >> 16) & 0x8000u);
    u &= 0x7fffffffu;
    if (u > 0x7f800000u) return uint16_t(sign | 0x7e00u);   // NaN
    if (u >= 0x477ff000u) return uint16_t(sign | (u >= 0x7f800000u ? 0x7c00u : 0x7bffu));  // clamp to 65504 (inf stays inf)
    if (u < 0x33000001u) return sign;                       // |f| <= 2^-25 rounds to zero
    const int e = int(u >> 23) - 127;                       // unbiased exponent
    uint32_t m = (u & 0x7fffffu) | 0x800000u;               // 24-bit significand
    int shift;                                              // bits dropped from m
    uint32_t base;
    if (e >= -14) {
        shift = 13;
        base = uint32_t(e + 15) << 10;                      // (the hidden bit of m carries into the exponent field)
        m &= 0x7fffffu;
    } else {
        shift = 13 + (-14 - e);                             // subnormal: value = m * 2^(e-23) in units of 2^-24
        base = 0;
    }
    const uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    uint32_t r = base + q;
    if (rem > half || (rem == half && (q & 1u))) r++;       // round to nearest even (may carry into the exponent)
    return uint16_t(sign | r);
}

float f16_to_f32(uint16_t h) {
    const uint32_t sign = uint32_t(h & 0x8000u) << 16;
    const int e = (h >> 10) & 31;
    const uint32_t m = h & 0x3ffu;
    float v;
    if (e == 0) v = std::ldexp(float(m), -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = std::ldexp(float(m | 0x400u), e - 25);
    uint32_t u;
    std::memcpy(&u, &v, 4);
    u |= sign;
    std::memcpy(&v, &u, 4);
    return v;
}

void split2h_host(float v, uint16_t p[3]) {
    if (v > 65504.f) v = 65504.f;
    if (v < -65504.f) v = -65504.f;
    p[0] = f16_rne(v);
    p[1] = f16_rne(v - f16_to_f32(p[0]));
    p[2] = 0;
}

void set_sx_f16(bool on) { t_sx_f16 = on; }
void set_sx_h1(bool on) { t_sx_h1 = on; }
void set_sx_force16(bool on) { t_sx_force16 = on; }
void set_sx_shape32(bool on) { t_sx_shape32 = on; }

// Arithmetic of the split-operand convs for the opens that follow on this thread: an explicit choice
// (vits_open_opts) wins over VITSMI_GEN_PRECISION in the environment; nullptr / "" = the default (f16x3).
thread_local std::string t_precision_override;
thread_local bool t_precision_set = false;
void set_gen_precision_override(const char *name) {
    t_precision_set = name != nullptr;
    t_precision_override = name ? name : "";
}
const char *gen_precision_name() {
    return t_precision_set ? t_precision_override.c_str() : std::getenv("VITSMI_GEN_PRECISION");
}

bool sx_supported(int Cin, int Cout_virtual, int Cr, int K, int dil) {
    if (Cin < 16 || Cin % 16 || Cout_virtual % 32 || Cr % 32 || K < 1 || dil < 1) return false;
    const int cfg = sx_pick_cfg(Cout_virtual);
    // an x stage is at most 12 DMA rounds of 4 KiB (conv_sx_engine.hip.hpp launch_conv_sx)
    const size_t LW = size_t(sx_tile_n(cfg)) + size_t((K < 3 ? 3 : K) - 1) * dil;
    const size_t x_bytes = (6 * LW * 16 + 4095) / 4096 * 4096;
    if (sx_raw_format(Cin) && 2 * LW > 768) return false;  // raw-input staging: three cells per thread
    return x_bytes <= 12 * 4096;
}

void set_tiling_override(int cfg, int ck) {
    t_cfg_override = cfg;
    t_ck_override = ck;
}

std::string pack_test_conv(const float *w, const float *bias, int Cin, int Cout, int K, int dil, int pad_l, int hint,
                           ConvDesc *d, std::vector<float> *arena) {
    if (Cin < 1 || Cout < 1 || K < 1 || dil < 1 || pad_l < 0) return "bad conv shape";
    t_hint = hint;
    Packer P(*arena);
    P.alloc(256);  // zero page at offset 0
    auto wf = [&](int co, int ci, int tap) { return w[(int64_t(co) * Cin + ci) * K + tap]; };
    try {
        if (hint == 3) *d = pack_conv_sx(P, Cin, Cout, K, dil, pad_l, wf, bias);
        else *d = pack_conv(P, Cin, Cout, K, dil, pad_l, wf, bias);
    } catch (const std::exception &e) {
        return e.what();
    }
    return "";
}

std::string pack_test_convT(const float *w, const float *bias, int Cin, int Cout, int K, int stride, ConvDesc *d,
                            std::vector<float> *arena, bool sx) {
    try {
        t_hint = 0;
        Resolver R;
        TRef t;
        t.p = w;
        t.dims = {Cin, Cout, K};
        R.t["t.weight"] = t;
        if (bias) {
            TRef b;
            b.p = bias;
            b.dims = {Cout};
            R.t["t.bias"] = b;
        }
        R.ints["t.stride"] = stride;
        R.ints["t.pad"] = (K - stride) / 2;
        Packer P(*arena);
        P.alloc(256);  // zero page at offset 0
        *d = sx ? pack_convT_sx(P, R, "t") : pack_convT(P, R, "t");
    } catch (const std::exception &e) {
        return e.what();
    }
    return "";
}

std::string Model::build(const OnnxModel &om, bool layout_only) {
    t_sx_f16 = false;
    t_sx_h1 = false;
    t_sx_force16 = false;
    try {
        Resolver R;
        resolve(om, R);
        arena.clear();
        Packer P(arena, layout_only);
        zeros_off = P.alloc(1024);  // zero page: padding source of the conv engine's LDS-DMA
        input_names = om.inputs;
        meta = om.meta;
        // (not from the file: what the loader has to report about it; absent when there is nothing to report)
        if (!R.name_warnings.empty()) meta["vitsmi.name_warnings"] = R.name_warnings;
        // Graph inputs (voice.py:347-373 filters its feed by these names).  "langid" appears in third-party
        // multi-lingual exports (voice.py:369): it is accepted when nothing in the graph consumes a language table
        // this engine would have to apply; anything else is not a graph this engine understands.
        for (const auto &n : input_names)
            if (n != "input" && n != "input_lengths" && n != "scales" && n != "sid" && n != "langid")
                throw std::runtime_error("unsupported graph input '" + n + "'");
        for (const auto &kv : R.t)
            if (kv.first.find("emb_l") != std::string::npos || kv.first.find("emb_lang") != std::string::npos)
                throw std::runtime_error("language-embedding conditioned voices are not supported (" + kv.first + ")");

        // ---------------- text encoder (models.py:168-209, attentions.py)
        t_hint = 2;
        const TRef &embw = R.need("enc_p.emb.weight", 2);
        n_vocab = int(embw.dims[0]);
        H = int(embw.dims[1]);
        emb = P.put(embw);
        // The encoder's convs once more for the split-operand engine (same arithmetic as the generator's default: two fp16
        // planes, three products, 16x16x32 loop): 3-5 x the f32 matrix cores' rate at batch 32, and a shorter dependent
        // chain per workgroup at batch 1.  Taken when the generator runs f16x3 (the exact mode keeps the f32 engine here:
        // it is what a range-guard fallback falls back to); VITSMI_ENC_ENGINE=f32 for A/B timing.
        bool enc_want_sx = false;
        {
            const char *ge = std::getenv("VITSMI_GEN_ENGINE"), *ee = std::getenv("VITSMI_ENC_ENGINE");
            const char *pe = gen_precision_name();
            enc_want_sx = !(ge && std::string(ge) == "f32") && !(ee && std::string(ee) == "f32") &&
                          precision_is_fp16_family(pe) && H % 32 == 0;
        }
        bool enc_all_sx = enc_want_sx;
        auto sx_twin = [&](const ConvDesc &f, const std::function<ConvDesc()> &mk) {
            ConvDesc d;
            if (!enc_want_sx || f.Cin % 32 || f.Cout % 32 || !sx_supported(f.Cin, f.Cout, f.Cout, f.K, f.dil)) {
                enc_all_sx = false;
                return d;
            }
            t_sx_f16 = true;
            d = mk();
            t_sx_f16 = false;
            if (!d.sx || !d.f16 || !d.s16 || d.K != f.K) enc_all_sx = false;
            return d;
        };
        for (int l = 0;; l++) {
            std::string a = "enc_p.encoder.attn_layers." + std::to_string(l);
            if (!R.get(a + ".conv_q.weight")) break;
            EncLayerDesc L;
            const int64_t HH = int64_t(H) * H;
            const TRef &wq = R.need(a + ".conv_q.weight", 3, HH), &wk = R.need(a + ".conv_k.weight", 3, HH),
                       &wv = R.need(a + ".conv_v.weight", 3, HH);
            const TRef &bq = R.need(a + ".conv_q.bias", 1, H), &bk = R.need(a + ".conv_k.bias", 1, H),
                       &bv = R.need(a + ".conv_v.bias", 1, H);
            if (wq.dims[0] != H || wq.dims[2] != 1 || wk.dims[0] != H || wv.dims[0] != H)
                throw std::runtime_error("attention projections must be 1x1 convs of the hidden width");
            // fused q|k|v projection: one [3H, H, 1] conv
            const float *ws[3] = {wq.p, wk.p, wv.p};
            int Hh = H;
            auto wf = [&](int co, int ci, int) { return ws[co / Hh][int64_t(co % Hh) * Hh + ci]; };
            std::vector<float> b3(size_t(3) * H);
            std::memcpy(b3.data(), bq.p, size_t(H) * 4);
            std::memcpy(b3.data() + H, bk.p, size_t(H) * 4);
            std::memcpy(b3.data() + 2 * H, bv.p, size_t(H) * 4);
            L.qkv = pack_conv(P, H, 3 * H, 1, 1, 0, wf, b3.data());
            L.o = pack_named(P, R, a + ".conv_o", 1, 0);
            L.qkv_sx = sx_twin(L.qkv, [&] { return pack_conv_sx(P, H, 3 * H, 1, 1, 0, wf, b3.data()); });
            L.o_sx = sx_twin(L.o, [&] { return pack_named_sx(P, R, a + ".conv_o", 1, 0); });
            const TRef &rk = R.need(a + ".emb_rel_k", 3);
            if (rk.dims[0] != 1) throw std::runtime_error("per-head relative embeddings are unsupported");
            if (rk.dims[1] % 2 != 1) throw std::runtime_error(a + ".emb_rel_k: expected 2*window+1 rows");
            const int dk_l = int(rk.dims[2]), window_l = int(rk.dims[1] - 1) / 2;
            if (dk_l <= 0 || H % dk_l) throw std::runtime_error(a + ": head width does not divide the hidden width");
            if (l > 0 && (dk_l != dk || window_l != window)) throw std::runtime_error(a + ": attention layers differ in shape");
            const TRef &rv = R.need(a + ".emb_rel_v", 3, rk.numel());
            window = window_l;
            dk = dk_l;
            n_heads = H / dk;
            L.rel_k = P.put(rk);
            L.rel_v = P.put(rv);
            L.ln1_g = P.put(R.need("enc_p.encoder.norm_layers_1." + std::to_string(l) + ".gamma", 1, H));
            L.ln1_b = P.put(R.need("enc_p.encoder.norm_layers_1." + std::to_string(l) + ".beta", 1, H));
            L.ln2_g = P.put(R.need("enc_p.encoder.norm_layers_2." + std::to_string(l) + ".gamma", 1, H));
            L.ln2_b = P.put(R.need("enc_p.encoder.norm_layers_2." + std::to_string(l) + ".beta", 1, H));
            if (L.o.Cin != H || L.o.Cout != H) throw std::runtime_error(a + ".conv_o: unexpected shape");
            std::string f = "enc_p.encoder.ffn_layers." + std::to_string(l);
            const TRef &w1 = R.need(f + ".conv_1.weight", 3);
            int fk = int(w1.dims[2]);
            FF = int(w1.dims[0]);
            L.ffn1 = pack_named(P, R, f + ".conv_1", 1, (fk - 1) / 2);  // attentions.py:419-427
            L.ffn2 = pack_named(P, R, f + ".conv_2", 1, (fk - 1) / 2);
            if (L.ffn1.Cin != H || L.ffn2.Cin != FF || L.ffn2.Cout != H) throw std::runtime_error(f + ": unexpected FFN shape");
            L.ffn1_sx = sx_twin(L.ffn1, [&] { return pack_named_sx(P, R, f + ".conv_1", 1, (fk - 1) / 2); });
            L.ffn2_sx = sx_twin(L.ffn2, [&] { return pack_named_sx(P, R, f + ".conv_2", 1, (fk - 1) / 2); });
            enc.push_back(L);
        }
        n_layers = int(enc.size());
        if (!n_layers) throw std::runtime_error("no encoder attention layers found");
        enc_proj = pack_named(P, R, "enc_p.proj", 1, 0);
        if (enc_proj.Cin != H || enc_proj.K != 1 || enc_proj.Cout % 2) throw std::runtime_error("enc_p.proj: unexpected shape");
        C = enc_proj.Cout / 2;
        enc_proj_sx = sx_twin(enc_proj, [&] { return pack_named_sx(P, R, "enc_p.proj", 1, 0); });
        enc_sx = enc_all_sx;

        // ---------------- speaker embedding (models.py:614-615)
        if (R.get("emb_g.weight")) {
            const TRef *eg = &R.need("emb_g.weight", 2);
            n_speakers = int(eg->dims[0]);
            gin = int(eg->dims[1]);
            emb_g = P.put(*eg);
        }

        // ---------------- duration predictor
        use_sdp = R.get("dp.flows.0.m") != nullptr;
        if (use_sdp) {
            dp_pre = pack_named(P, R, "dp.pre", 1, 0);
            dp_proj = pack_named(P, R, "dp.proj", 1, 0);
            dp_proj16 = pack_pw16(P, R, "dp.proj", dp_proj.Cin);
            dp_convs = pack_dds(P, R, "dp.convs");
            int order[3] = {7, 5, 3};  // models.py:109-110
            for (int i = 0; i < 3; i++) {
                std::string s = "dp.flows." + std::to_string(order[i]);
                const int Cd = dp_pre.Cout;
                cf[i].pre_w = P.put(R.need(s + ".pre.weight", 3, Cd));  // Conv1d(1, C, 1)
                cf[i].pre_b = P.put(R.need(s + ".pre.bias", 1, Cd));
                cf[i].convs = pack_dds(P, R, s + ".convs");
                cf[i].proj = pack_named(P, R, s + ".proj", 1, 0);
                cf[i].proj16 = pack_pw16(P, R, s + ".proj", cf[i].proj.Cin);
                if (cf[i].proj.Cin != Cd || cf[i].convs.l[0].pw.Cin != Cd || (cf[i].proj.Cout + 1) % 3)
                    throw std::runtime_error(s + ": unexpected ConvFlow shape");
                cf[i].nb = (cf[i].proj.Cout + 1) / 3;
                if (cf[i].nb < 1 || cf[i].nb > 16) throw std::runtime_error("spline with more than 16 bins is unsupported");
            }
            if (dp_pre.Cin != H || dp_proj.Cin != dp_pre.Cout || dp_proj.Cout != dp_pre.Cout ||
                dp_convs.l[0].pw.Cin != dp_pre.Cout)
                throw std::runtime_error("dp: unexpected stochastic duration predictor shape");
            if (R.req("dp.flows.0.m").numel() < 1 || R.req("dp.flows.0.logs").numel() < 1)
                throw std::runtime_error("dp.flows.0: empty ElementwiseAffine parameters");
            ea_m0 = R.req("dp.flows.0.m").p[0];
            ea_logs0 = R.req("dp.flows.0.logs").p[0];
        } else {
            const TRef &w1 = R.need("dp.conv_1.weight", 3);
            int k = int(w1.dims[2]);
            dpp_F = int(w1.dims[0]);
            dpp_conv1 = pack_named(P, R, "dp.conv_1", 1, k / 2);  // models.py:138-144
            dpp_conv2 = pack_named(P, R, "dp.conv_2", 1, k / 2);
            dpp_proj = pack_named(P, R, "dp.proj", 1, 0);
            dpp_n1_g = P.put(R.need("dp.norm_1.gamma", 1, dpp_F));
            dpp_n1_b = P.put(R.need("dp.norm_1.beta", 1, dpp_F));
            dpp_n2_g = P.put(R.need("dp.norm_2.gamma", 1, dpp_F));
            dpp_n2_b = P.put(R.need("dp.norm_2.beta", 1, dpp_F));
            if (dpp_conv1.Cin != H || dpp_conv2.Cin != dpp_F || dpp_conv2.Cout != dpp_F || dpp_proj.Cin != dpp_F ||
                dpp_proj.Cout != 1)
                throw std::runtime_error("dp: unexpected duration predictor shape");
        }
        if (gin) {
            const TRef &cw = R.need("dp.cond.weight", 3);
            dp_cond_rows = int(cw.dims[0]);
            if (cw.dims[1] != gin || cw.dims[2] != 1 || dp_cond_rows != (use_sdp ? dp_pre.Cout : H))
                throw std::runtime_error("dp.cond: unexpected shape");
            dp_cond_w = P.put(cw);
            dp_cond_b = P.put(R.need("dp.cond.bias", 1, dp_cond_rows));
        }

        // ---------------- flow (models.py:212-254), Flip folded into channel permutations
        t_hint = 1;
        {
            int nfl = 0;
            while (R.get("flow.flows." + std::to_string(2 * nfl) + ".pre.weight")) nfl++;
            if (!nfl) throw std::runtime_error("no coupling layers found");
            int half = C / 2;
            std::vector<int> rev(half);
            for (int i = 0; i < half; i++) rev[i] = half - 1 - i;
            for (int e = 0; e < nfl; e++) {
                int idx = 2 * (nfl - 1 - e);
                std::string s = "flow.flows." + std::to_string(idx);
                CouplingDesc cd;
                cd.swapped = ((e + 1) % 2) == 1;  // odd number of Flips so far
                cd.pre = pack_named(P, R, s + ".pre", 1, 0, cd.swapped ? &rev : nullptr, nullptr);
                cd.post = pack_named(P, R, s + ".post", 1, 0, nullptr, cd.swapped ? &rev : nullptr);
                if (cd.post.Cout != half) throw std::runtime_error("only mean_only coupling layers are supported");
                if (cd.pre.Cin != half || cd.pre.K != 1 || cd.post.K != 1 || cd.post.Cin != cd.pre.Cout ||
                    (e > 0 && cd.pre.Cout != flow_H))
                    throw std::runtime_error(s + ": unexpected coupling layer shape");
                flow_H = cd.pre.Cout;
                for (int i = 0; i < 8; i++) {
                    std::string in = s + ".enc.in_layers." + std::to_string(i);
                    if (!R.get(in + ".weight")) break;
                    const TRef *w = &R.need(in + ".weight", 3);
                    if (w->dims[0] != 2 * flow_H || w->dims[1] != flow_H) throw std::runtime_error(in + ": unexpected WN shape");
                    int k = int(w->dims[2]);
                    int dil = int(R.geti(in + ".dilation", 1));
                    if (dil < 1) throw std::runtime_error(in + ": bad dilation");
                    // the WN dilated conv carries 83 % of the flow's FLOPs: split-exact engine when the shape allows
                    // (plane input, 128/64-row tiles); VITSMI_GEN_ENGINE=f32 keeps everything on the f32 engine
                    const char *env = std::getenv("VITSMI_GEN_ENGINE");
                    const bool f32_only = env && std::string(env) == "f32";
                    const int ci = int(w->dims[1]), co = int(w->dims[0]);
                    if (!f32_only && !sx_raw_format(ci) && co % 64 == 0 && ci % 8 == 0 && sx_supported(ci, co, co, k, dil)) {
                        const char *pe = gen_precision_name();  // (same arithmetic as the generator)
                        t_sx_f16 = precision_is_fp16_family(pe);
                        // frame-domain tensors are short (F ~ 3 T): 64-row tiles give the grid twice the workgroups
                        // (288 -> 576 at batch 32), measured 3.04 -> 2.87 ms for the flow
                        t_sx_min_cfg = 1;
                        // Gate folded into this conv's epilogue (SX_GATE): rows permuted so that every 64-row tile holds
                        // 32 tanh channels and their 32 sigmoid partners (row r of tile m: channel 32 m + r for r < 32,
                        // H + 32 m + r - 32 above).  VITSMI_FLOW_NO_GATE keeps the separate gate kernel (A/B timing).
                        static const bool no_gate = std::getenv("VITSMI_FLOW_NO_GATE") != nullptr;
                        std::vector<int> perm;
                        const bool gate = !no_gate && co == 2 * flow_H && flow_H % 32 == 0 && sx_pick_cfg(co) == 1;
                        if (gate) {
                            perm.resize(co);
                            for (int r = 0; r < co; r++) {
                                const int m = r / 64, rr = r % 64;
                                perm[r] = rr < 32 ? 32 * m + rr : flow_H + 32 * m + rr - 32;
                            }
                        }
                        cd.wn[i].in = pack_named_sx(P, R, in, dil, same_pad(k, dil), gate ? &perm : nullptr);
                        cd.wn[i].in.gate = gate;
                        t_sx_min_cfg = 0;
                        t_sx_f16 = false;
                    } else
                        cd.wn[i].in = pack_named(P, R, in, dil, same_pad(k, dil));
                    cd.wn[i].rs = pack_named(P, R, s + ".enc.res_skip_layers." + std::to_string(i), 1, 0);
                    {
                        // ... and for the split-operand engine, where the gated in-layer runs on its 16x16x32 loop: the 1 x 1
                        // conv then reads the acts as operand planes and folds the x / skip update into its epilogue
                        // (SX_WN_RMW).  VITSMI_FLOW_RS_F32 keeps it on the f32 engine (A/B timing).
                        static const bool rs_f32 = std::getenv("VITSMI_FLOW_RS_F32") != nullptr;
                        const auto &inl = cd.wn[i].in;
                        if (!rs_f32 && inl.sx && inl.f16 && inl.gate && inl.s16 && flow_H % 32 == 0) {
                            t_sx_f16 = true;
                            ConvDesc r2 = pack_named_sx(P, R, s + ".enc.res_skip_layers." + std::to_string(i), 1, 0);
                            t_sx_f16 = false;
                            if (r2.s16 && r2.K == 1) cd.wn[i].rs_sx = r2;
                        }
                    }
                    if (cd.wn[i].rs.Cin != flow_H || cd.wn[i].rs.K != 1 ||
                        (cd.wn[i].rs.Cout != flow_H && cd.wn[i].rs.Cout != 2 * flow_H))
                        throw std::runtime_error(s + ": unexpected res_skip shape");
                    cd.n_wn = i + 1;
                }
                if (!cd.n_wn || cd.wn[cd.n_wn - 1].rs.Cout != flow_H)
                    throw std::runtime_error(s + ": WN stack must end in a skip-only layer");
                for (int i = 0; i + 1 < cd.n_wn; i++)
                    if (cd.wn[i].rs.Cout != 2 * flow_H) throw std::runtime_error(s + ": unexpected res_skip shape");
                {
                    // pre / post on the split-operand engine too when the whole WN stack runs there: pre reads the planes
                    // the previous coupling's post wrote (x1 of one coupling is x0 of the next), post reads the planes of
                    // the skip sum the last res_skip conv wrote.  VITSMI_FLOW_PREPOST_F32 keeps the f32 engine (A/B timing).
                    static const bool pp_f32 = std::getenv("VITSMI_FLOW_PREPOST_F32") != nullptr;
                    bool all_sx = !pp_f32 && half % 32 == 0 && flow_H % 32 == 0;
                    for (int i = 0; i < cd.n_wn; i++)
                        all_sx = all_sx && cd.wn[i].in.sx && cd.wn[i].in.f16 && cd.wn[i].in.gate && cd.wn[i].in.s16 && cd.wn[i].rs_sx.sx;
                    if (all_sx && sx_supported(half, flow_H, flow_H, 1, 1) && sx_supported(flow_H, half, half, 1, 1)) {
                        const TRef &wpre = R.need(s + ".pre.weight", 3), &wpost = R.need(s + ".post.weight", 3);
                        const TRef *bpre = R.bias_of(s + ".pre.bias", flow_H), *bpost = R.bias_of(s + ".post.bias", half);
                        const bool sw = cd.swapped;
                        auto wf_pre = [&](int co, int ci, int) { return wpre.p[int64_t(co) * half + (sw ? rev[ci] : ci)]; };
                        auto wf_post = [&](int co, int ci, int) { return wpost.p[int64_t(sw ? rev[co] : co) * flow_H + ci]; };
                        std::vector<float> bperm;
                        const float *bp = bpost ? bpost->p : nullptr;
                        if (bpost && sw) {
                            bperm.resize(half);
                            for (int c = 0; c < half; c++) bperm[c] = bpost->p[rev[c]];
                            bp = bperm.data();
                        }
                        t_sx_f16 = true;
                        ConvDesc a = pack_conv_sx(P, half, flow_H, 1, 1, 0, wf_pre, bpre ? bpre->p : nullptr);
                        ConvDesc b = pack_conv_sx(P, flow_H, half, 1, 1, 0, wf_post, bp);
                        t_sx_f16 = false;
                        if (a.s16 && b.s16 && a.K == 1 && b.K == 1) {
                            cd.pre_sx = a;
                            cd.post_sx = b;
                        }
                    }
                }
                if (gin) {
                    const int64_t rows = int64_t(2) * flow_H * cd.n_wn;
                    cd.cond_w = P.put(R.need(s + ".enc.cond_layer.weight", 3, rows * gin));
                    cd.cond_b = P.put(R.need(s + ".enc.cond_layer.bias", 1, rows));
                }
                flow.push_back(cd);
            }
            if (nfl % 2) throw std::runtime_error("odd number of coupling layers is unsupported");
        }

        // ---------------- generator (models.py:299-368)
        t_hint = 0;
        int nups = 0;
        while (R.get("dec.ups." + std::to_string(nups) + ".weight")) nups++;
        int nrb = 0;
        while (R.get("dec.resblocks." + std::to_string(nrb) + ".convs1.0.weight") ||
               R.get("dec.resblocks." + std::to_string(nrb) + ".convs.0.weight"))
            nrb++;
        if (!nups || nrb % nups) throw std::runtime_error("unexpected generator structure");
        const int nk = nrb / nups;
        // Engine choice for the whole generator (its tensors change layout with the engine): split-operand
        // when every conv qualifies, else the f32 engine.  VITSMI_GEN_ENGINE=f32 forces the latter (A/B runs).
        {
            const char *env = std::getenv("VITSMI_GEN_ENGINE");
            bool ok = !(env && std::string(env) == "f32");
            auto conv_ok = [&](const std::string &name) {
                const TRef *w = R.get(name + ".weight");
                if (!w || w->dims.size() != 3) return false;
                const int k = int(w->dims[2]), dil = int(R.geti(name + ".dilation", 1));
                return sx_supported(int(w->dims[1]), int(w->dims[0]), int(w->dims[0]), k, dil);
            };
            ok = ok && conv_ok("dec.conv_pre");
            for (int i = 0; ok && i < nups; i++) {
                const ConvTGeom g = convt_geom(R, "dec.ups." + std::to_string(i));
                ok = sx_supported(g.Cin, g.Cout * g.u, g.Cout, g.o_max - g.o_min + 1, 1);
            }
            for (int j = 0; ok && j < nrb; j++)
                for (int q = 0; ok && q < 4; q++) {
                    const std::string rb = "dec.resblocks." + std::to_string(j);
                    const bool t1 = R.get(rb + ".convs1.0.weight") != nullptr;
                    const std::string c1 = rb + (t1 ? ".convs1." : ".convs.") + std::to_string(q);
                    if (!R.get(c1 + ".weight")) break;
                    ok = conv_ok(c1) && (!t1 || conv_ok(rb + ".convs2." + std::to_string(q)));
                }
            gen_sx = ok;
            // default: the generator's convs on two fp16 planes / three products per fp32 product (fp32-grade error);
            // VITSMI_GEN_PRECISION=bf16x6 (exact products, six bf16 plane products), bf16x3 or bf16 pack bf16 planes
            const char *pe = gen_precision_name();
            gen_f16 = gen_sx && (!pe || !*pe || std::string(pe) == "f16x3");
            // "f16": ONE fp16 plane per operand, one product, activations stored as fp16 (BASELINE config 4's reduced-
            // precision vocoder); needs every generator conv on the 16x16x32 loop (pack_conv_sx throws where it cannot)
            gen_h1 = gen_sx && pe && std::string(pe) == "f16";
            if (pe && *pe && std::string(pe) != "f16x3" && std::string(pe) != "f16" && std::string(pe) != "bf16x6")
                throw std::runtime_error(std::string("unknown generator precision '") + pe + "' (VITSMI_GEN_PRECISION: f16x3, bf16x6, f16)");
        }
        t_sx_f16 = gen_f16;
        t_sx_h1 = gen_h1;
        // Both fp16 arithmetics run the PLANE-STREAM generator (vitsmi.hip run_generator_planes): every tensor between two
        // convs is stored once, as the operand planes of its consumer - also on the 32- / 64-channel stages, whose convs are
        // therefore packed for plane input on the 16x16x32 loop like everyone else (no raw-input kernels)
        // (measured r04l: in f16x3 the plane-stream form is 3-5 % SLOWER end to end - its fused pair kernel on the 16x16x32 loop
        // takes 1.35 / 0.85 ms per 64- / 32-channel ResBlock1 step against 1.14 / 0.72 for the 32x32x16 one on fp32 raw
        // tensors - so f16x3 keeps the raw-stream generator; VITSMI_F16X3_STREAM=planes selects the other for A/B runs)
        const char *se = std::getenv("VITSMI_F16X3_STREAM");
        gen_planes = gen_h1 || (gen_f16 && se && std::string(se) == "planes");
        t_sx_force16 = gen_f16 && gen_planes;
        auto gconv = [&](const std::string &name, int dil, int padL) {
            return gen_sx ? pack_named_sx(P, R, name, dil, padL) : pack_named(P, R, name, dil, padL);
        };
        {
            const TRef &wpre = R.need("dec.conv_pre.weight", 3);
            if (wpre.dims[1] != C) throw std::runtime_error("dec.conv_pre: input width differs from the flow's");
            conv_pre = gconv("dec.conv_pre", 1, int(wpre.dims[2] - 1) / 2);
        }
        C0 = conv_pre.Cout;
        if (gin) {
            dec_cond_w = P.put(R.need("dec.cond.weight", 3, int64_t(C0) * gin));
            dec_cond_b = P.put(R.need("dec.cond.bias", 1, C0));
        }
        hop = 1;
        for (int i = 0; i < nups; i++) {
            UpStageDesc st;
            st.up = gen_sx ? pack_convT_sx(P, R, "dec.ups." + std::to_string(i)) : pack_convT(P, R, "dec.ups." + std::to_string(i));
            st.u = st.up.ups;
            st.C = st.up.Cout / st.u;
            if (st.up.Cin != (i == 0 ? C0 : ups.back().C)) throw std::runtime_error("dec.ups: channel chain is broken");
            hop *= st.u;
            for (int j = 0; j < nk; j++) {
                std::string rb = "dec.resblocks." + std::to_string(i * nk + j);
                ResBlockDesc rd;
                rd.type1 = R.get(rb + ".convs1.0.weight") != nullptr;
                for (int q = 0; q < 4; q++) {
                    std::string c1 = rb + (rd.type1 ? ".convs1." : ".convs.") + std::to_string(q);
                    if (!R.get(c1 + ".weight")) break;
                    const TRef *w = &R.need(c1 + ".weight", 3);
                    int k = int(w->dims[2]);
                    int dil = int(R.geti(c1 + ".dilation", 1));
                    if (dil < 1 || w->dims[0] != st.C || w->dims[1] != st.C) throw std::runtime_error(c1 + ": unexpected shape");
                    rd.c1[q] = gconv(c1, dil, same_pad(k, dil));
                    if (rd.type1) {
                        std::string c2 = rb + ".convs2." + std::to_string(q);
                        const TRef &w2 = R.need(c2 + ".weight", 3);
                        int k2 = int(w2.dims[2]);
                        int d2 = int(R.geti(c2 + ".dilation", 1));
                        if (d2 < 1 || w2.dims[0] != st.C || w2.dims[1] != st.C) throw std::runtime_error(c2 + ": unexpected shape");
                        rd.c2[q] = gconv(c2, d2, same_pad(k2, d2));
                    }
                    rd.n = q + 1;
                }
                st.rbs.push_back(rd);
            }
            ups.push_back(st);
        }
        t_sx_f16 = false;
        t_sx_h1 = false;
        t_sx_force16 = false;
        const TRef &pw = R.need("dec.conv_post.weight", 3);
        if (pw.dims[0] != 1) throw std::runtime_error("conv_post must have one output channel");
        if (ups.empty() || pw.dims[1] != ups.back().C) throw std::runtime_error("conv_post: input width differs from the last stage");
        if (R.get("dec.conv_post.bias")) throw std::runtime_error("conv_post with bias is unsupported");
        post_cin = int(pw.dims[1]);
        post_k = int(pw.dims[2]);
        post_w = P.put(pw);

        // ---------------- reference-definition work counts
        {
            double t = 1, macs = conv_pre.macs_per_t, el = double(conv_pre.Cin) + conv_pre.Cout;
            for (auto &st : ups) {
                macs += t * st.up.macs_per_t;
                el += t * st.up.Cin;
                t *= st.u;
                el += t * st.C;
                for (auto &rb : st.rbs)
                    for (int q = 0; q < rb.n; q++) {
                        macs += t * rb.c1[q].macs_per_t;
                        el += t * 2 * st.C;
                        if (rb.type1) {
                            macs += t * rb.c2[q].macs_per_t;
                            el += t * 2 * st.C;
                        }
                    }
            }
            macs += t * post_cin * post_k;
            el += t * (post_cin + 1);
            dec_macs_per_frame = macs;
            dec_elems_per_frame = el;
            double fm = 0;
            for (auto &cd : flow) {
                fm += cd.pre.macs_per_t + cd.post.macs_per_t;
                for (int i = 0; i < cd.n_wn; i++) fm += cd.wn[i].in.macs_per_t + cd.wn[i].rs.macs_per_t;
            }
            flow_macs_per_frame = fm;
            double em = enc_proj.macs_per_t;
            for (auto &L : enc) em += L.qkv.macs_per_t + L.o.macs_per_t + L.ffn1.macs_per_t + L.ffn2.macs_per_t;
            enc_macs_per_token = em;  // attention contraction added at run time (depends on T)
        }
        // One-sided receptive field of the generator in input frames (chunked rendering discards this much on either
        // side of a chunk; models.py:348-368): conv_pre, then per stage the transposed conv's reach (in its input steps)
        // and the widest ResBlock of the multi-receptive-field bank, each divided by the cumulative upsampling.
        {
            double r = double(conv_pre.K - 1 - conv_pre.padL > conv_pre.padL ? conv_pre.K - 1 - conv_pre.padL : conv_pre.padL);
            double rate = 1;
            std::vector<double> part;  // reach of each stage (its transposed conv + its widest ResBlock), in frames
            for (auto &st : ups) {
                const int right = st.up.K - 1 - st.up.padL;  // dense-conv form of the transposed conv: taps -padL .. right
                double p = double(right > st.up.padL ? right : st.up.padL) / rate;
                rate *= st.u;
                double widest = 0;
                for (auto &rb : st.rbs) {
                    double w = 0;
                    for (int q = 0; q < rb.n; q++) {
                        // (sx packing pads narrow kernels to 3 taps with zeros: padL is still the real left reach)
                        w += double(rb.c1[q].padL);
                        if (rb.type1) w += double(rb.c2[q].padL);
                    }
                    widest = w > widest ? w : widest;
                }
                p += widest / rate;
                part.push_back(p);
                r += p;
            }
            const double tail = double(post_k / 2) / rate;
            r += tail;
            gen_rf_frames = int(std::ceil(r)) + 1;
            // ... and what is left of it from the input of stage s on (ragged rendering: the launches of stage s end an
            // utterance's tensors this many frames behind its end - every launch of a stage the same, they share the running sum)
            gen_rf_stage.assign(ups.size(), 0);
            double rest = tail;
            for (int si = int(ups.size()) - 1; si >= 0; si--) {
                rest += part[size_t(si)];
                gen_rf_stage[size_t(si)] = int(std::ceil(rest)) + 1;
            }
        }
        if (input_names.empty()) {
            input_names = {"input", "input_lengths", "scales"};
            if (gin) input_names.push_back("sid");
        }
        arena_floats = P.size;
    } catch (const std::exception &e) {
        return e.what();
    }
    return "";
}


// ====================================================================================== ByT5 G2P
// Parameters are found by following nodes, as for VITS: every Linear is a MatMul whose second input is an anonymous
// transposed initializer [in, out]; node names carry the module path (".../encoder/block.3/layer.0/SelfAttention/q/MatMul",
// with whatever prefix the exporting wrapper added).

namespace {

struct T5Refs {
    std::map<std::string, TRef> t;  // "encoder.3.0.SelfAttention.q" -> [in, out]; "...layer_norm" -> [d]; "shared"; "lm_head"
};

// module path components after the last "encoder/" or "decoder/" component of a node name
bool t5_parse(const std::string &name, std::string &stack, std::vector<std::string> &rest) {
    std::vector<std::string> parts;
    std::string cur;
    for (char c : name) {
        if (c == '/') {
            if (!cur.empty()) parts.push_back(cur);
            cur.clear();
        } else
            cur.push_back(c);
    }
    if (!cur.empty()) parts.push_back(cur);
    int at = -1;
    for (int i = 0; i < int(parts.size()); i++)
        if (parts[i] == "encoder" || parts[i] == "decoder") at = i;
    if (at < 0) return false;
    stack = parts[at];
    rest.assign(parts.begin() + at + 1, parts.end());
    return true;
}

int t5_index(const std::string &s, const char *prefix) {  // "block.12" -> 12
    const size_t n = std::strlen(prefix);
    if (s.compare(0, n, prefix) != 0 || s.size() == n) return -1;
    int v = 0;
    for (size_t i = n; i < s.size(); i++) {
        if (s[i] < '0' || s[i] > '9') return -1;
        v = v * 10 + (s[i] - '0');
        if (v > 4096) return -1;
    }
    return v;
}

}  // namespace

std::string G2PModel::build(const OnnxModel &om) {
    try {
        T5Refs R;
        auto init = [&](const std::string &n) -> const OnnxTensor * { return om.find_init(n); };
        auto put = [&](const std::string &key, const OnnxTensor *ot, size_t rank) {
            if (!ot || !ot->data() || ot->dims.size() != rank || R.t.count(key)) return;
            for (auto d : ot->dims)
                if (d <= 0) return;
            const float *p = ot->data();
            const int64_t n = ot->numel();
            for (int64_t i = 0; i < n; i++)
                if (!std::isfinite(p[i])) throw std::runtime_error("non-finite value in parameter " + key);
            TRef r;
            r.p = p;
            r.dims = ot->dims;
            R.t[key] = r;
        };
        std::map<std::string, const OnnxNode *> producer;
        const OnnxNode *lm_node = nullptr;
        bool saw_tanh = false, saw_erf = false, saw_relu = false;
        for (const auto &n : om.nodes) {
            for (const auto &o : n.outputs) producer[o] = &n;
            if (n.name.empty()) continue;
            if (n.name.find("DenseReluDense") != std::string::npos) {
                saw_tanh = saw_tanh || n.op == "Tanh";
                saw_erf = saw_erf || n.op == "Erf";
                saw_relu = saw_relu || n.op == "Relu";
            }
            if (n.op == "MatMul" && n.inputs.size() > 1 && n.name.size() >= 14 &&
                n.name.compare(n.name.size() - 14, 14, "lm_head/MatMul") == 0) {
                put("lm_head", init(n.inputs[1]), 2);
                lm_node = &n;
                continue;
            }
            std::string stack;
            std::vector<std::string> rest;
            if (!t5_parse(n.name, stack, rest)) continue;
            if (n.op == "Gather" && rest.size() == 2 && rest[0] == "embed_tokens" && !n.inputs.empty()) {
                put("shared", init(n.inputs[0]), 2);
                continue;
            }
            if (n.op == "Mul" && rest.size() == 2 && rest[0] == "final_layer_norm") {
                for (const auto &i : n.inputs) put(stack + ".final_layer_norm", init(i), 1);
                continue;
            }
            if (rest.size() < 3) continue;
            const int blk = t5_index(rest[0], "block."), lay = t5_index(rest[1], "layer.");
            if (blk < 0 || lay < 0) continue;
            const std::string base = stack + "." + std::to_string(blk) + "." + std::to_string(lay);
            if (n.op == "Mul" && rest.size() == 4 && rest[2] == "layer_norm") {
                for (const auto &i : n.inputs) put(base + ".layer_norm", init(i), 1);
            } else if (n.op == "MatMul" && rest.size() == 5 && n.inputs.size() > 1) {
                put(base + "." + rest[2] + "." + rest[3], init(n.inputs[1]), 2);  // SelfAttention.q, DenseReluDense.wi_0 ...
            } else if (n.op == "Gather" && rest.size() == 5 && rest[3] == "relative_attention_bias" && !n.inputs.empty()) {
                put(base + "." + rest[2] + ".relative_attention_bias", init(n.inputs[0]), 2);
            }
        }
        auto need = [&](const std::string &k, size_t rank) -> const TRef & {
            auto it = R.t.find(k);
            if (it == R.t.end()) throw std::runtime_error("T5 parameter not found in graph: " + k);
            if (it->second.dims.size() != rank) throw std::runtime_error(k + ": unexpected rank");
            return it->second;
        };
        input_names = om.inputs;
        output_names = om.outputs;
        for (const auto &n : input_names)
            if (n != "input_ids" && n != "attention_mask" && n != "decoder_input_ids")
                throw std::runtime_error("unsupported graph input '" + n + "' (expected the T5 seq2seq signature)");
        const TRef &emb = need("shared", 2);
        vocab = int(emb.dims[0]);
        d_model = int(emb.dims[1]);
        const TRef &rb = need("encoder.0.0.SelfAttention.relative_attention_bias", 2);
        num_buckets = int(rb.dims[0]);
        heads = int(rb.dims[1]);
        const TRef &q0 = need("encoder.0.0.SelfAttention.q", 2);
        if (q0.dims[0] != d_model || heads <= 0 || q0.dims[1] % heads) throw std::runtime_error("unexpected attention shape");
        inner = int(q0.dims[1]);
        d_kv = inner / heads;
        // (num_buckets >= 4: the bidirectional table halves it twice - max_exact = num_buckets / 4 must not be zero)
        if (d_kv > 256 || num_buckets < 4 || num_buckets % 2 || num_buckets / 2 >= max_distance) throw std::runtime_error("unsupported T5 attention geometry");
        act = saw_tanh ? 0 : (saw_erf ? 2 : (saw_relu ? 1 : 0));
        if (lm_node) {  // a Mul between the decoder's final layer norm and lm_head = the tied-embedding output scale
            auto it = producer.find(lm_node->inputs[0]);
            scale_out = it != producer.end() && it->second->op == "Mul" &&
                        it->second->name.find("final_layer_norm") == std::string::npos;
        }
        arena.clear();
        Packer P(arena, false);
        zeros_off = P.alloc(1024);
        auto linear = [&](const std::string &key, int in, int out) {
            const TRef &w = need(key, 2);
            if (w.dims[0] != in || w.dims[1] != out)
                throw std::runtime_error(key + ": expected [" + std::to_string(in) + ", " + std::to_string(out) + "]");
            const float *wp = w.p;
            const int64_t ld = w.dims[1];
            T5Linear L;
            L.in = in;
            L.out = out;
            L.rowmajor = P.alloc(int64_t(in) * out);
            float *dst = P.arena.data() + L.rowmajor;
            for (int ci0 = 0; ci0 < in; ci0 += 64)  // (blocked transpose: MatMul initializers are [in, out])
                for (int co = 0; co < out; co++)
                    for (int ci = ci0; ci < in && ci < ci0 + 64; ci++) dst[int64_t(co) * in + ci] = wp[int64_t(ci) * ld + co];
            return L;
        };
        auto vec = [&](const std::string &key) {
            const TRef &g = need(key, 1);
            if (g.dims[0] != d_model) throw std::runtime_error(key + ": expected d_model values");
            return P.put(g);
        };
        shared = P.put(emb);
        enc_bias = P.put(rb);
        const TRef &rbd = need("decoder.0.0.SelfAttention.relative_attention_bias", 2);
        if (rbd.dims[0] != num_buckets || rbd.dims[1] != heads) throw std::runtime_error("decoder bias table differs from the encoder's");
        dec_bias = P.put(rbd);
        auto attn = [&](const std::string &pfx) {
            T5AttnDesc a;
            a.q = linear(pfx + ".q", d_model, inner);
            a.k = linear(pfx + ".k", d_model, inner);
            a.v = linear(pfx + ".v", d_model, inner);
            a.o = linear(pfx + ".o", inner, d_model);
            return a;
        };
        auto ffn = [&](const std::string &pfx) {
            T5FfnDesc f;
            f.gated = R.t.count(pfx + ".wi_0") != 0;
            const TRef &w0 = need(pfx + (f.gated ? ".wi_0" : ".wi"), 2);
            const int ff = int(w0.dims[1]);
            if (d_ff && ff != d_ff) throw std::runtime_error("feed-forward widths differ between blocks");
            d_ff = ff;
            f.wi0 = linear(pfx + (f.gated ? ".wi_0" : ".wi"), d_model, ff);
            if (f.gated) f.wi1 = linear(pfx + ".wi_1", d_model, ff);
            f.wo = linear(pfx + ".wo", ff, d_model);
            return f;
        };
        for (int b = 0;; b++) {
            const std::string e = "encoder." + std::to_string(b);
            if (!R.t.count(e + ".0.SelfAttention.q")) break;
            T5BlockDesc d;
            d.ln_self = vec(e + ".0.layer_norm");
            d.self = attn(e + ".0.SelfAttention");
            d.ln_ffn = vec(e + ".1.layer_norm");
            d.ffn = ffn(e + ".1.DenseReluDense");
            enc.push_back(d);
        }
        for (int b = 0;; b++) {
            const std::string e = "decoder." + std::to_string(b);
            if (!R.t.count(e + ".0.SelfAttention.q")) break;
            T5BlockDesc d;
            d.ln_self = vec(e + ".0.layer_norm");
            d.self = attn(e + ".0.SelfAttention");
            d.ln_cross = vec(e + ".1.layer_norm");
            d.cross = attn(e + ".1.EncDecAttention");
            d.ln_ffn = vec(e + ".2.layer_norm");
            d.ffn = ffn(e + ".2.DenseReluDense");
            dec.push_back(d);
        }
        if (enc.empty() || dec.empty()) throw std::runtime_error("no T5 encoder / decoder blocks found");
        enc_final_ln = vec("encoder.final_layer_norm");
        dec_final_ln = vec("decoder.final_layer_norm");
        lm_head = linear("lm_head", d_model, vocab);
        arena_floats = P.size;
        // relative-position buckets (integers: the float32 arithmetic of the published function, step by step)
        bucket_enc.assign(2 * kMaxPos - 1, 0);
        bucket_dec.assign(2 * kMaxPos - 1, 0);
        for (int d = -(kMaxPos - 1); d < kMaxPos; d++) {
            for (int bidir = 0; bidir < 2; bidir++) {
                int nb = num_buckets, ret = 0, n;
                if (bidir) {
                    nb /= 2;
                    ret = d > 0 ? nb : 0;
                    n = d < 0 ? -d : d;
                } else
                    n = d < 0 ? -d : 0;
                const int max_exact = nb / 2;
                int bucket;
                if (n < max_exact) bucket = n;
                else {
                    const float ratio = float(n) / float(max_exact);
                    const float denom = float(std::log(double(max_distance) / double(max_exact)));  // python float -> f32 scalar
                    const float v = std::log(ratio) / denom * float(nb - max_exact);
                    bucket = max_exact + int(v);
                    if (bucket > nb - 1) bucket = nb - 1;
                }
                (bidir ? bucket_enc : bucket_dec)[d + kMaxPos - 1] = ret + bucket;
            }
        }
    } catch (const std::exception &e) {
        return e.what();
    }
    return "";
}

}  // namespace vitsmi
