// onnx_import.cpp - ONNX model files -> hp_model (layer list + weight blob), the job nvonnxparser does for the
// reference's dnn::tensorrt(const onnx&, ...) constructor (include/hyperpose/operator/dnn/tensorrt.hpp:53-62,
// src/tensorrt.cpp:162-223: parseFromFile, exactly one input, 3 channels, H x W from the caller, dynamic batch).
//
// No protobuf / onnx library: an ONNX file is a protobuf-wire ModelProto and the subset read here (graph, nodes,
// attributes, initializers, value infos) is a page of varint / length-delimited decoding, written against the public
// onnx.proto3 field numbers.  The graph is then LOWERED onto the three layer kinds the engine runs:
//   Conv (group 1 / depthwise) ............ HP_OP_CONV / HP_OP_DWCONV, ONNX `pads` kept (or recognised as TF "SAME")
//   BatchNormalization, per-channel Mul/Add/Sub/Div with constants ... folded into the producing convolution
//   Relu, Clip(0,6), LeakyRelu, PRelu ..... fused as the producing layer's activation
//   Sigmoid, Softplus ..................... output post-ops (only on graph outputs, as the engine evaluates them in fp32 there)
//   Add of two maps ....................... the later convolution's residual input
//   Concat(axis 1) ........................ producers retargeted to write at a channel offset of one tensor (no copy)
//   Resize / Upsample by an integer factor (nearest, or linear with half-pixel centres) ... HP_OP_UPSAMPLE
//   MaxPool, Pad (merged into its consumer), Identity / Dropout, scalar or per-channel arithmetic on the input image
//   (folded into the engine's mean / inv_std), Transpose NHWC->NCHW directly on the input.
// Whatever does not fit a fused form falls back to an identity 1x1 convolution carrying the activation / residual /
// copy, so the lowering stays general; operators outside this list fail with the node name and operator in hp_last_error().
#include "model.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------- protobuf wire format
struct pb {
    const uint8_t *p, *end;
    bool ok = true;
    bool more() const { return ok && p < end; }
    uint64_t varint()
    {
        uint64_t v = 0;
        for (int shift = 0; shift < 64; shift += 7) {
            if (p >= end) {
                ok = false;
                return 0;
            }
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80))
                return v;
        }
        ok = false;
        return 0;
    }
    bool tag(uint32_t& field, uint32_t& wt)
    {
        if (!more())
            return false;
        const uint64_t t = varint();
        field = (uint32_t)(t >> 3), wt = (uint32_t)(t & 7);
        return ok;
    }
    pb sub()
    {
        const uint64_t n = varint();
        if (!ok || n > (uint64_t)(end - p)) {
            ok = false;
            return pb{ p, p };
        }
        pb r{ p, p + n };
        p += n;
        return r;
    }
    std::string str()
    {
        pb s = sub();
        return std::string((const char*)s.p, (size_t)(s.end - s.p));
    }
    uint32_t fixed32()
    {
        if (end - p < 4) {
            ok = false;
            return 0;
        }
        uint32_t v;
        memcpy(&v, p, 4);
        p += 4;
        return v;
    }
    uint64_t fixed64()
    {
        if (end - p < 8) {
            ok = false;
            return 0;
        }
        uint64_t v;
        memcpy(&v, p, 8);
        p += 8;
        return v;
    }
    void skip(uint32_t wt)
    {
        switch (wt) {
        case 0: (void)varint(); break;
        case 1: (void)fixed64(); break;
        case 2: (void)sub(); break;
        case 5: (void)fixed32(); break;
        default: ok = false;
        }
    }
};

float half_to_float(uint16_t h)
{
    const uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
    uint32_t u;
    if (e == 0) {
        if (m == 0)
            u = s << 31;
        else {
            int ee = -1;
            uint32_t mm = m;
            do {
                ++ee;
                mm <<= 1;
            } while (!(mm & 1024));
            u = (s << 31) | ((uint32_t)(127 - 15 - ee) << 23) | ((mm & 1023) << 13);
        }
    } else if (e == 31)
        u = (s << 31) | 0x7f800000u | (m << 13);
    else
        u = (s << 31) | ((e + 127 - 15) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// ------------------------------------------------------------------------------------------- the ONNX subset
struct o_tensor {
    std::string name;
    std::vector<int64_t> dims;
    int dtype = 0;
    std::vector<float> f;   // every numeric type, as float
    std::vector<int64_t> i; // integer types, exact
    bool external = false;
    // number of elements; SIZE_MAX when a dimension is negative or the product does not fit (never equal to a real payload size)
    size_t count() const
    {
        size_t n = 1;
        for (int64_t d : dims) {
            if (d < 0)
                return SIZE_MAX;
            if (d != 0 && n > (SIZE_MAX >> 1) / (size_t)d)
                return SIZE_MAX;
            n *= (size_t)d;
        }
        return n;
    }
};
struct o_attr {
    std::string name;
    float f = 0;
    int64_t i = 0;
    std::string s;
    std::vector<float> floats;
    std::vector<int64_t> ints;
    std::shared_ptr<o_tensor> t;
};
struct o_node {
    std::string op, name;
    std::vector<std::string> in, out;
    std::vector<o_attr> attrs;
    const o_attr* attr(const char* n) const
    {
        for (const auto& a : attrs)
            if (a.name == n)
                return &a;
        return nullptr;
    }
    int64_t geti(const char* n, int64_t dflt) const
    {
        const o_attr* a = attr(n);
        return a ? a->i : dflt;
    }
    float getf(const char* n, float dflt) const
    {
        const o_attr* a = attr(n);
        return a ? a->f : dflt;
    }
};
struct o_value_info {
    std::string name;
    std::vector<int64_t> dims; // -1 = symbolic / unknown
    bool has_shape = false;
};
struct o_graph {
    std::string name;
    std::vector<o_node> nodes;
    std::map<std::string, std::shared_ptr<o_tensor>> init;
    std::vector<o_value_info> inputs, outputs;
    int64_t opset = 0;
};

bool parse_tensor(pb r, o_tensor& t)
{
    std::vector<float> fdata;
    std::vector<double> ddata;
    std::vector<int64_t> idata;
    std::string raw;
    bool has_raw = false;
    uint32_t f, wt;
    while (r.tag(f, wt)) {
        if (f == 1) { // dims
            if (wt == 2) {
                pb s = r.sub();
                while (s.more())
                    t.dims.push_back((int64_t)s.varint());
            } else
                t.dims.push_back((int64_t)r.varint());
        } else if (f == 2)
            t.dtype = (int)r.varint();
        else if (f == 4) { // float_data
            if (wt == 2) {
                pb s = r.sub();
                while (s.more()) {
                    const uint32_t u = s.fixed32();
                    float v;
                    memcpy(&v, &u, 4);
                    fdata.push_back(v);
                }
            } else {
                const uint32_t u = r.fixed32();
                float v;
                memcpy(&v, &u, 4);
                fdata.push_back(v);
            }
        } else if (f == 5 || f == 7) { // int32_data / int64_data
            if (wt == 2) {
                pb s = r.sub();
                while (s.more())
                    idata.push_back((int64_t)s.varint());
            } else
                idata.push_back((int64_t)r.varint());
        } else if (f == 10) { // double_data
            if (wt == 2) {
                pb s = r.sub();
                while (s.more()) {
                    const uint64_t u = s.fixed64();
                    double v;
                    memcpy(&v, &u, 8);
                    ddata.push_back(v);
                }
            } else {
                const uint64_t u = r.fixed64();
                double v;
                memcpy(&v, &u, 8);
                ddata.push_back(v);
            }
        } else if (f == 8)
            t.name = r.str();
        else if (f == 9) {
            raw = r.str();
            has_raw = true;
        } else if (f == 14) {
            if (r.varint() == 1)
                t.external = true;
        } else
            r.skip(wt);
    }
    if (!r.ok)
        return false;
    // the file is untrusted input: dimensions are bounded before anything is sized or indexed by them
    if (t.dims.size() > 8)
        return false;
    for (int64_t d : t.dims)
        if (d < 0 || d > 0x7fffffff)
            return false;
    const size_t n = t.count();
    if (n == SIZE_MAX)
        return false;
    auto from_raw = [&](size_t esize, auto conv) {
        if (raw.size() % esize != 0 || raw.size() / esize != n) // (not n * esize: that product can wrap)
            return false;
        for (size_t k = 0; k < n; ++k)
            conv(raw.data() + k * esize);
        return true;
    };
    switch (t.dtype) {
    case 1: // FLOAT
        if (has_raw)
            return from_raw(4, [&](const char* q) { float v; memcpy(&v, q, 4); t.f.push_back(v); });
        t.f = fdata;
        return t.f.size() == n;
    case 11: // DOUBLE
        if (has_raw)
            return from_raw(8, [&](const char* q) { double v; memcpy(&v, q, 8); t.f.push_back((float)v); });
        for (double v : ddata)
            t.f.push_back((float)v);
        return t.f.size() == n;
    case 10: // FLOAT16 (int32_data carries the bit patterns)
        if (has_raw)
            return from_raw(2, [&](const char* q) { uint16_t v; memcpy(&v, q, 2); t.f.push_back(half_to_float(v)); });
        for (int64_t v : idata)
            t.f.push_back(half_to_float((uint16_t)v));
        return t.f.size() == n;
    case 7: // INT64
        if (has_raw) {
            if (!from_raw(8, [&](const char* q) { int64_t v; memcpy(&v, q, 8); t.i.push_back(v); }))
                return false;
        } else
            t.i = idata;
        break;
    case 6: // INT32
        if (has_raw) {
            if (!from_raw(4, [&](const char* q) { int32_t v; memcpy(&v, q, 4); t.i.push_back(v); }))
                return false;
        } else
            for (int64_t v : idata)
                t.i.push_back((int32_t)v);
        break;
    case 2: case 3: case 9: // UINT8 / INT8 / BOOL
        if (has_raw) {
            if (!from_raw(1, [&](const char* q) { t.i.push_back(t.dtype == 3 ? (int64_t)(int8_t)*q : (int64_t)(uint8_t)*q); }))
                return false;
        } else
            t.i = idata;
        break;
    default:
        return t.external; // other element types are only tolerated when nothing reads them
    }
    for (int64_t v : t.i)
        t.f.push_back((float)v);
    return t.i.size() == n;
}

bool parse_attr(pb r, o_attr& a)
{
    uint32_t f, wt;
    while (r.tag(f, wt)) {
        if (f == 1)
            a.name = r.str();
        else if (f == 2) {
            const uint32_t u = r.fixed32();
            memcpy(&a.f, &u, 4);
        } else if (f == 3)
            a.i = (int64_t)r.varint();
        else if (f == 4)
            a.s = r.str();
        else if (f == 5) {
            a.t = std::make_shared<o_tensor>();
            if (!parse_tensor(r.sub(), *a.t))
                return false;
        } else if (f == 7) {
            if (wt == 2) {
                pb s = r.sub();
                while (s.more()) {
                    const uint32_t u = s.fixed32();
                    float v;
                    memcpy(&v, &u, 4);
                    a.floats.push_back(v);
                }
            } else {
                const uint32_t u = r.fixed32();
                float v;
                memcpy(&v, &u, 4);
                a.floats.push_back(v);
            }
        } else if (f == 8) {
            if (wt == 2) {
                pb s = r.sub();
                while (s.more())
                    a.ints.push_back((int64_t)s.varint());
            } else
                a.ints.push_back((int64_t)r.varint());
        } else
            r.skip(wt);
    }
    return r.ok;
}

bool parse_node(pb r, o_node& n)
{
    uint32_t f, wt;
    while (r.tag(f, wt)) {
        if (f == 1)
            n.in.push_back(r.str());
        else if (f == 2)
            n.out.push_back(r.str());
        else if (f == 3)
            n.name = r.str();
        else if (f == 4)
            n.op = r.str();
        else if (f == 5) {
            o_attr a;
            if (!parse_attr(r.sub(), a))
                return false;
            n.attrs.push_back(std::move(a));
        } else
            r.skip(wt);
    }
    return r.ok;
}

bool parse_value_info(pb r, o_value_info& v)
{
    uint32_t f, wt;
    while (r.tag(f, wt)) {
        if (f == 1)
            v.name = r.str();
        else if (f == 2) { // TypeProto
            pb ty = r.sub();
            uint32_t f2, w2;
            while (ty.tag(f2, w2)) {
                if (f2 != 1) { // tensor_type
                    ty.skip(w2);
                    continue;
                }
                pb tt = ty.sub();
                uint32_t f3, w3;
                while (tt.tag(f3, w3)) {
                    if (f3 != 2) { // shape
                        tt.skip(w3);
                        continue;
                    }
                    v.has_shape = true;
                    pb sh = tt.sub();
                    uint32_t f4, w4;
                    while (sh.tag(f4, w4)) {
                        if (f4 != 1) {
                            sh.skip(w4);
                            continue;
                        }
                        pb dim = sh.sub();
                        int64_t value = -1;
                        uint32_t f5, w5;
                        while (dim.tag(f5, w5)) {
                            if (f5 == 1)
                                value = (int64_t)dim.varint();
                            else
                                dim.skip(w5);
                        }
                        v.dims.push_back(value);
                        if (!dim.ok)
                            return false;
                    }
                    if (!sh.ok)
                        return false;
                }
                if (!tt.ok)
                    return false;
            }
            if (!ty.ok)
                return false;
        } else
            r.skip(wt);
    }
    return r.ok;
}

bool parse_graph(pb r, o_graph& g, std::string& why)
{
    uint32_t f, wt;
    while (r.tag(f, wt)) {
        if (f == 1) {
            o_node n;
            if (!parse_node(r.sub(), n)) {
                why = "malformed NodeProto";
                return false;
            }
            g.nodes.push_back(std::move(n));
        } else if (f == 2)
            g.name = r.str();
        else if (f == 5) {
            auto t = std::make_shared<o_tensor>();
            if (!parse_tensor(r.sub(), *t)) {
                why = "initializer '" + t->name + "': malformed or unsupported element type";
                return false;
            }
            if (t->external) {
                why = "initializer '" + t->name + "' is stored in an external data file";
                return false;
            }
            g.init[t->name] = t;
        } else if (f == 11 || f == 12) {
            o_value_info v;
            if (!parse_value_info(r.sub(), v)) {
                why = "malformed ValueInfoProto";
                return false;
            }
            (f == 11 ? g.inputs : g.outputs).push_back(std::move(v));
        } else
            r.skip(wt);
    }
    if (!r.ok)
        why = "malformed GraphProto";
    return r.ok;
}

bool parse_model(const uint8_t* data, size_t size, o_graph& g, std::string& why)
{
    pb r{ data, data + size };
    uint32_t f, wt;
    bool have_graph = false;
    while (r.tag(f, wt)) {
        if (f == 7) {
            if (!parse_graph(r.sub(), g, why))
                return false;
            have_graph = true;
        } else if (f == 8) { // opset_import
            pb s = r.sub();
            std::string domain;
            int64_t version = 0;
            uint32_t f2, w2;
            while (s.tag(f2, w2)) {
                if (f2 == 1)
                    domain = s.str();
                else if (f2 == 2)
                    version = (int64_t)s.varint();
                else
                    s.skip(w2);
            }
            if (domain.empty() || domain == "ai.onnx")
                g.opset = version;
        } else
            r.skip(wt);
    }
    if (!r.ok || !have_graph) {
        if (why.empty())
            why = "not an ONNX ModelProto (no graph)";
        return false;
    }
    return true;
}

// ------------------------------------------------------------------------------------------- lowering
struct lower_error {
    std::string msg;
};
std::string printable(std::string s) // names come from the file: keep error messages plain ASCII and short
{
    if (s.size() > 80)
        s = s.substr(0, 77) + "...";
    for (char& c : s)
        if (c < 0x20 || c > 0x7e)
            c = '?';
    return s;
}
[[noreturn]] void fail(const o_node* n, const std::string& what)
{
    if (n)
        throw lower_error{ printable("onnx: node '" + (n->name.empty() ? (n->out.empty() ? std::string("?") : n->out[0]) : n->name) + "' (" + n->op + "): ") + printable(what) };
    throw lower_error{ "onnx: " + printable(what) };
}

struct val {
    enum { NONE, IMAGE, MAP, CONST } kind = NONE;
    int tensor = -1, coff = 0, C = 0, H = 0, W = 0;
    int producer = -1;    // the one layer that wrote exactly this value, or -1
    int post_act = 0;     // Sigmoid / Softplus waiting for the output conversion
    // ... and the coordinate arithmetic the PoseProposal export ends in (hyperpose/Model/pose_proposal/model.py:111-119, restore_coor:
    // x = (sigmoid + grid_x) * 32, w = sigmoid * 384): after the activation the engine adds the pixel's column (1) / row (2) index and
    // multiplies by a scalar (hp_output_desc::grid / scale)
    int post_grid = 0;
    float post_scale = 1.f;
    int pad[4] = { 0, 0, 0, 0 }; // a Pad node waiting for its consumer (top, left, bottom, right)
    bool nhwc = false;    // IMAGE declared as N,H,W,3 and not yet transposed
    // post-processing views of a feature map (the PoseProposal / PifPaf exports end in Split / Reshape / Transpose nodes):
    std::vector<int64_t> view; // non-empty: a Reshape of the SAME row-major memory, may only reach a graph output or another Reshape
    bool nhwc_view = false;    // a MAP transposed N,C,H,W -> N,H,W,C that must be transposed back before anything reads it
    float a[3] = { 1, 1, 1 }, b[3] = { 0, 0, 0 }; // IMAGE: y = a * x + b so far
    std::shared_ptr<o_tensor> c;
};

struct lowering {
    const o_graph& g;
    hp_model& m;
    std::vector<float>& blob;
    std::map<std::string, val> vals;
    std::map<std::string, int> uses;      // consumers of every value (node inputs + graph outputs)
    std::map<std::string, int> remaining; // ... that have not been lowered yet
    std::map<int, int> tensor_c;      // channels of each tensor
    std::map<int, int> last_writer;   // last layer writing into each tensor
    std::set<int> in_concat;          // tensors that are concat targets (or members moved into one)

    lowering(const o_graph& g_, hp_model& m_) : g(g_), m(m_), blob(m_.weights) {}

    val& get(const o_node& n, size_t k)
    {
        if (k >= n.in.size() || n.in[k].empty())
            fail(&n, "missing input " + std::to_string(k));
        auto it = vals.find(n.in[k]);
        if (it == vals.end()) {
            auto ci = g.init.find(n.in[k]);
            if (ci == g.init.end())
                fail(&n, "input '" + n.in[k] + "' is not produced by any node");
            val v;
            v.kind = val::CONST, v.c = ci->second;
            it = vals.emplace(n.in[k], v).first;
        }
        return it->second;
    }
    bool has_in(const o_node& n, size_t k) const { return k < n.in.size() && !n.in[k].empty(); }

    int64_t append(const float* p, size_t n)
    {
        const int64_t off = (int64_t)blob.size();
        blob.insert(blob.end(), p, p + n);
        return off;
    }

    int emit(const hp_layer& L)
    {
        m.layers.push_back(L);
        m.init_scale.push_back(1.f), m.init_bias.push_back(0.f);
        const int idx = (int)m.layers.size() - 1;
        last_writer[L.out] = idx;
        tensor_c[L.out] = std::max(tensor_c[L.out], L.out_coff + L.cout);
        return idx;
    }

    static void same_pads(int in, int k, int stride, int dil, int& before, int& after)
    {
        const int out = (in + stride - 1) / stride;
        const int total = std::max((out - 1) * stride + (k - 1) * dil + 1 - in, 0);
        before = total / 2, after = total - before;
    }
    // write the padding of a Conv / MaxPool node into L (recognising TF "SAME") and return the output size
    void set_geometry(const o_node& n, const val& x, hp_layer& L, int& OH, int& OW)
    {
        int pads[4] = { 0, 0, 0, 0 };
        const o_attr* pa = n.attr("pads");
        const o_attr* ap = n.attr("auto_pad");
        const std::string mode = ap ? ap->s : "NOTSET";
        if (mode == "SAME_UPPER" || mode == "SAME_LOWER") {
            int b, a;
            same_pads(x.H, L.kh, L.stride, L.dil, b, a);
            pads[0] = mode == "SAME_UPPER" ? b : a, pads[2] = mode == "SAME_UPPER" ? a : b;
            same_pads(x.W, L.kw, L.stride, L.dil, b, a);
            pads[1] = mode == "SAME_UPPER" ? b : a, pads[3] = mode == "SAME_UPPER" ? a : b;
        } else if (mode == "VALID" || mode == "NOTSET") {
            if (pa && mode == "NOTSET") {
                if (pa->ints.size() != 4)
                    fail(&n, "only 2-D pads are supported");
                for (int k = 0; k < 4; ++k)
                    pads[k] = (int)pa->ints[k];
            }
        } else
            fail(&n, "auto_pad '" + mode + "'");
        for (int k = 0; k < 4; ++k)
            pads[k] += x.pad[k];
        if (n.geti("ceil_mode", 0) != 0)
            fail(&n, "ceil_mode = 1");
        int sb, sa, tb, ta;
        same_pads(x.H, L.kh, L.stride, L.dil, sb, sa);
        same_pads(x.W, L.kw, L.stride, L.dil, tb, ta);
        if (pads[0] == sb && pads[2] == sa && pads[1] == tb && pads[3] == ta)
            L.pad_explicit = 0;
        else {
            L.pad_explicit = 1;
            for (int k = 0; k < 4; ++k)
                L.pad[k] = pads[k];
        }
        hp_layer_out_size(L, x.H, x.W, OH, OW);
        if (OH < 1 || OW < 1)
            fail(&n, "empty output");
    }
    void square_attr(const o_node& n, const char* name, int dflt, int& v)
    {
        const o_attr* a = n.attr(name);
        v = dflt;
        if (!a)
            return;
        if (a->ints.size() != 2 || a->ints[0] != a->ints[1])
            fail(&n, std::string(name) + " must be 2-D and equal in both dimensions");
        v = (int)a->ints[0];
    }

    // a constant as one value per channel of a C-channel map (scalar, [C], [C,1,1], [1,C,1,1])
    std::vector<float> per_channel(const o_node& n, const val& c, int C)
    {
        const o_tensor& t = *c.c;
        const size_t cnt = t.count();
        if (cnt == 1)
            return std::vector<float>((size_t)C, t.f.at(0));
        bool ok = cnt == (size_t)C;
        if (ok) { // exactly one non-unit dimension, and it is the channel axis of a [.., C, 1, 1] or [C] shape
            const size_t nd = t.dims.size();
            if (nd == 1)
                ok = true;
            else if (nd == 3)
                ok = t.dims[0] == C;
            else if (nd == 4)
                ok = t.dims[1] == C;
            else
                ok = false;
        }
        if (!ok)
            fail(&n, "constant operand is neither a scalar nor per-channel");
        return t.f;
    }

    // scale / shift the output channels of layer p: y' = s * y + t (valid only while the layer has no activation / residual)
    void fold_affine(int p, const std::vector<float>& s, const std::vector<float>& t)
    {
        hp_layer& L = m.layers[p];
        const size_t per = L.op == HP_OP_CONV ? (size_t)L.kh * L.kw * L.cin : (size_t)L.kh * L.kw;
        for (int c = 0; c < L.cout; ++c)
            for (size_t k = 0; k < per; ++k)
                blob[L.w_off + c * per + k] *= s[c];
        if (L.b_off < 0) {
            std::vector<float> zeros((size_t)L.cout, 0.f);
            L.b_off = append(zeros.data(), zeros.size());
        }
        for (int c = 0; c < L.cout; ++c)
            blob[L.b_off + c] = blob[L.b_off + c] * s[c] + t[c];
    }
    bool foldable(const std::string& name, const val& v) const
    {
        if (v.kind != val::MAP || v.producer < 0 || v.post_act)
            return false;
        const hp_layer& L = m.layers[v.producer];
        auto u = uses.find(name);
        return L.op != HP_OP_MAXPOOL && L.op != HP_OP_UPSAMPLE && L.act == HP_ACT_NONE && L.res < 0 && u != uses.end() && u->second == 1;
    }

    // identity 1x1 convolution reading v: the general carrier for an activation, a residual add or a copy
    int identity(const o_node& n, const val& v, int out_tensor, int out_coff)
    {
        if (v.coff % 8)
            fail(&n, "needs a copy of a map that starts at channel " + std::to_string(v.coff) + " of a concatenation (not 8-aligned)");
        hp_layer L;
        memset(&L, 0, sizeof(L));
        L.op = HP_OP_CONV, L.in = v.tensor, L.in_coff = v.coff, L.res = -1, L.cin = L.cout = v.C;
        L.out = out_tensor < 0 ? m.new_tensor() : out_tensor, L.out_coff = out_coff;
        L.kh = L.kw = L.stride = L.dil = 1, L.act = HP_ACT_NONE, L.b_off = -1, L.alpha_off = -1;
        std::vector<float> eye((size_t)v.C * v.C, 0.f);
        for (int c = 0; c < v.C; ++c)
            eye[(size_t)c * v.C + c] = 1.f;
        L.w_off = append(eye.data(), eye.size());
        return emit(L);
    }
    val materialise(const o_node& n, const val& v)
    {
        const int p = identity(n, v, -1, 0);
        val r;
        r.kind = val::MAP, r.tensor = m.layers[p].out, r.coff = 0, r.C = v.C, r.H = v.H, r.W = v.W, r.producer = p;
        return r;
    }
    void need_map(const o_node& n, const val& v)
    {
        if (v.kind != val::MAP)
            fail(&n, v.kind == val::IMAGE ? "cannot be applied to the network input directly" : "expects a feature map, got a constant");
        if (v.post_act || v.post_grid || v.post_scale != 1.f)
            fail(&n, "reads the result of a Sigmoid / Softplus (supported on graph outputs only)");
        if (!v.view.empty() || v.nhwc_view)
            fail(&n, "reads a reshaped / transposed feature map (Reshape and Transpose are supported as output post-processing only)");
        if (v.pad[0] | v.pad[1] | v.pad[2] | v.pad[3])
            if (n.op != "Conv" && n.op != "MaxPool")
                fail(&n, "a Pad node must feed a Conv or MaxPool");
    }

    void activation(const o_node& n, int act, float param, const std::vector<float>* alpha)
    {
        val x = get(n, 0);
        // element-wise: the layout does not matter - exporters that keep TensorFlow's N,H,W,C order (tf2onnx) leave an activation between
        // a Transpose pair; the pending N,H,W,C flag travels through it and is cancelled by the closing Transpose
        const bool nhwc = x.kind == val::MAP && x.nhwc_view && x.view.empty();
        x.nhwc_view = nhwc ? false : x.nhwc_view;
        need_map(n, x);
        if (!foldable(n.in[0], x) && !(x.producer >= 0 && m.layers[x.producer].op == HP_OP_CONV && m.layers[x.producer].act == HP_ACT_NONE
                && m.layers[x.producer].in != 0 && uses[n.in[0]] == 1))
            x = materialise(n, x);
        if (act == HP_ACT_PRELU && m.layers[x.producer].in == 0)
            x = materialise(n, x); // the first layer (fused u8 -> f32 conversion) has no per-channel slopes
        hp_layer& L = m.layers[x.producer];
        L.act = act, L.act_param = param;
        if (alpha)
            L.alpha_off = append(alpha->data(), alpha->size());
        if (L.res >= 0)
            L.res_before_act = 1; // conv -> Add -> activation
        x.nhwc_view = nhwc;
        vals[n.out.at(0)] = x;
    }

    void conv(const o_node& n)
    {
        val x = get(n, 0);
        val w = get(n, 1);
        if (w.kind != val::CONST || w.c->dims.size() != 4)
            fail(&n, "weights must be a 4-D initializer");
        if (x.kind == val::IMAGE && x.nhwc)
            fail(&n, "the N,H,W,3 input must be transposed to N,3,H,W first");
        if (x.kind != val::IMAGE)
            need_map(n, x);
        const o_tensor& wt = *w.c;
        for (int k = 0; k < 4; ++k)
            if (wt.dims[k] < 1 || wt.dims[k] > (k < 2 ? 65536 : 31))
                fail(&n, "weight dimension " + std::to_string(k) + " = " + std::to_string(wt.dims[k]) + " is outside the supported range");
        if (wt.f.size() != wt.count())
            fail(&n, "weight initializer carries no data");
        const int cout = (int)wt.dims[0], cin_g = (int)wt.dims[1], kh = (int)wt.dims[2], kw = (int)wt.dims[3];
        const int group = (int)n.geti("group", 1);
        hp_layer L;
        memset(&L, 0, sizeof(L));
        L.in = x.kind == val::IMAGE ? 0 : x.tensor, L.in_coff = x.kind == val::IMAGE ? 0 : x.coff, L.res = -1;
        L.cin = x.C, L.cout = cout, L.kh = kh, L.kw = kw, L.act = HP_ACT_NONE, L.b_off = L.alpha_off = -1;
        square_attr(n, "strides", 1, L.stride);
        square_attr(n, "dilations", 1, L.dil);
        if (const o_attr* ks = n.attr("kernel_shape"))
            if (ks->ints.size() != 2 || ks->ints[0] != kh || ks->ints[1] != kw)
                fail(&n, "kernel_shape disagrees with the weight tensor");
        std::vector<float> packed;
        if (group == 1) {
            if (cin_g != x.C)
                fail(&n, "weights expect " + std::to_string(cin_g) + " input channels, the input has " + std::to_string(x.C));
            L.op = HP_OP_CONV;
            packed.resize(wt.f.size()); // [cout][cin][kh][kw] -> [cout][kh][kw][cin]
            for (int o = 0; o < cout; ++o)
                for (int c = 0; c < cin_g; ++c)
                    for (int t = 0; t < kh * kw; ++t)
                        packed[((size_t)o * kh * kw + t) * cin_g + c] = wt.f[((size_t)o * cin_g + c) * kh * kw + t];
        } else if (group == x.C && cin_g == 1 && cout == x.C) {
            L.op = HP_OP_DWCONV;
            packed = wt.f; // [c][1][kh][kw]
        } else
            fail(&n, "grouped convolution (group = " + std::to_string(group) + ") other than depthwise");
        int OH, OW;
        set_geometry(n, x, L, OH, OW);
        L.w_off = append(packed.data(), packed.size());
        if (has_in(n, 2)) {
            const val& b = get(n, 2);
            if (b.kind != val::CONST || b.c->count() != (size_t)cout)
                fail(&n, "bias must be an initializer of cout values");
            L.b_off = append(b.c->f.data(), b.c->f.size());
        }
        if (x.kind == val::IMAGE) { // the engine applies (x - mean) * inv_std in the first layer's load
            for (int c = 0; c < 3; ++c) {
                if (x.a[c] == 0.f)
                    fail(&n, "input scaling by zero");
                m.inv_std[c] = x.a[c], m.mean[c] = -x.b[c] / x.a[c];
            }
        }
        L.out = m.new_tensor(), L.out_coff = 0;
        val y;
        y.kind = val::MAP, y.tensor = L.out, y.C = cout, y.H = OH, y.W = OW;
        y.producer = emit(L);
        vals[n.out.at(0)] = y;
    }

    void maxpool(const o_node& n)
    {
        val x = get(n, 0);
        need_map(n, x);
        const o_attr* ks = n.attr("kernel_shape");
        if (!ks || ks->ints.size() != 2 || ks->ints[0] != ks->ints[1])
            fail(&n, "kernel_shape must be square");
        hp_layer L;
        memset(&L, 0, sizeof(L));
        L.op = HP_OP_MAXPOOL, L.in = x.tensor, L.in_coff = x.coff, L.res = -1, L.cin = L.cout = x.C;
        L.kh = L.kw = (int)ks->ints[0], L.dil = 1, L.w_off = L.b_off = L.alpha_off = -1;
        square_attr(n, "strides", 1, L.stride);
        int d;
        square_attr(n, "dilations", 1, d);
        if (d != 1)
            fail(&n, "dilated max-pool");
        int OH, OW;
        set_geometry(n, x, L, OH, OW);
        L.out = m.new_tensor();
        val y;
        y.kind = val::MAP, y.tensor = L.out, y.C = x.C, y.H = OH, y.W = OW;
        y.producer = emit(L);
        vals[n.out.at(0)] = y;
    }

    // Resize (opset >= 10) / Upsample (7-9) by an integer factor, equal in H and W: nearest (asymmetric + floor, what PyTorch and
    // tf2onnx emit) or linear with half-pixel centres
    void resize(const o_node& n)
    {
        val x = get(n, 0);
        need_map(n, x);
        const o_attr* ma = n.attr("mode");
        const std::string mode = ma ? ma->s : "nearest";
        const o_attr* ca = n.attr("coordinate_transformation_mode");
        const std::string ctm = ca ? ca->s : (n.op == "Upsample" ? "asymmetric" : "half_pixel");
        std::vector<float> scales;
        std::vector<int64_t> sizes;
        if (n.op == "Upsample") {
            if (const o_attr* a = n.attr("scales"))
                scales = a->floats;
            else if (has_in(n, 1) && get(n, 1).kind == val::CONST)
                scales = get(n, 1).c->f;
        } else {
            const size_t si = n.in.size() >= 3 ? 2 : 1; // opset 10: (X, scales); opset 11+: (X, roi, scales, sizes)
            if (has_in(n, si) && get(n, si).kind == val::CONST && get(n, si).c->count() > 0)
                scales = get(n, si).c->f;
            else if (has_in(n, 3) && get(n, 3).kind == val::CONST)
                sizes = get(n, 3).c->i;
        }
        int sc = 0;
        if (scales.size() == 4 && scales[0] == 1.f && scales[1] == 1.f && scales[2] == scales[3] && scales[2] == std::floor(scales[2]))
            sc = (int)scales[2];
        else if (sizes.size() == 4 && sizes[2] % x.H == 0 && sizes[3] % x.W == 0 && sizes[2] / x.H == sizes[3] / x.W && sizes[1] == x.C)
            sc = (int)(sizes[2] / x.H);
        if (sc < 1 || sc > 16)
            fail(&n, "only constant integer scales 1..16, equal in H and W, are supported");
        int kind;
        if (mode == "nearest") {
            if (ctm != "asymmetric")
                fail(&n, "nearest with coordinate_transformation_mode '" + ctm + "'");
            const o_attr* nm = n.attr("nearest_mode");
            if (nm && nm->s != "floor")
                fail(&n, "nearest_mode '" + nm->s + "'");
            kind = 0;
        } else if (mode == "linear") {
            if (ctm != "half_pixel" && ctm != "pytorch_half_pixel")
                fail(&n, "linear with coordinate_transformation_mode '" + ctm + "'");
            kind = 1;
        } else
            fail(&n, "mode '" + mode + "'");
        hp_layer L;
        memset(&L, 0, sizeof(L));
        L.op = HP_OP_UPSAMPLE, L.in = x.tensor, L.in_coff = x.coff, L.res = -1, L.cin = L.cout = x.C;
        L.kh = kind, L.kw = 1, L.stride = sc, L.dil = 1, L.w_off = L.b_off = L.alpha_off = -1;
        L.out = m.new_tensor();
        val y;
        y.kind = val::MAP, y.tensor = L.out, y.C = x.C, y.H = x.H * sc, y.W = x.W * sc;
        y.producer = emit(L);
        vals[n.out.at(0)] = y;
    }

    void batchnorm(const o_node& n)
    {
        const val x = get(n, 0);
        const int C = x.C;
        std::vector<float> p[4];
        for (int k = 0; k < 4; ++k) {
            const val& c = get(n, 1 + k);
            if (c.kind != val::CONST || c.c->count() != (size_t)C)
                fail(&n, "scale / bias / mean / var must be initializers of C values");
            p[k] = c.c->f;
        }
        const float eps = n.getf("epsilon", 1e-5f);
        std::vector<float> s((size_t)C), t((size_t)C);
        for (int c = 0; c < C; ++c) {
            s[c] = p[0][c] / std::sqrt(p[3][c] + eps);
            t[c] = p[1][c] - p[2][c] * s[c];
        }
        if (x.kind == val::IMAGE)
            fail(&n, "BatchNormalization directly on the input image");
        need_map(n, x);
        val y = foldable(n.in[0], x) ? x : materialise(n, x);
        fold_affine(y.producer, s, t);
        vals[n.out.at(0)] = y;
    }

    // Add / Sub / Mul / Div of a constant AFTER a Sigmoid / Softplus: only what the output conversion evaluates - a scalar factor, or the
    // addition of the cell-index grid (a constant [.., H, W] whose value is the column or the row index) before any factor
    bool post_arithmetic(const o_node& n, const val& x, const val& c, bool const_first)
    {
        if (x.kind != val::MAP || !(x.post_act || x.post_grid || x.post_scale != 1.f) || c.kind != val::CONST)
            return false;
        const o_tensor& t = *c.c;
        const size_t cnt = t.count();
        bool uniform = cnt >= 1;
        for (size_t k = 1; k < cnt && uniform; ++k)
            uniform = t.f[k] == t.f[0];
        val y = x;
        if (uniform && (n.op == "Mul" || (n.op == "Div" && !const_first))) {
            y.post_scale = n.op == "Mul" ? x.post_scale * t.f.at(0) : x.post_scale / t.f.at(0);
        } else if (n.op == "Add" && cnt == (size_t)x.H * x.W && x.post_grid == 0 && x.post_scale == 1.f && t.dims.size() >= 2
            && t.dims[t.dims.size() - 1] == x.W && t.dims[t.dims.size() - 2] == x.H) {
            bool is_x = true, is_y = true;
            for (int yy = 0; yy < x.H; ++yy)
                for (int xx = 0; xx < x.W; ++xx) {
                    const float v = t.f[(size_t)yy * x.W + xx];
                    is_x = is_x && v == (float)xx, is_y = is_y && v == (float)yy;
                }
            if (!is_x && !is_y)
                fail(&n, "a constant map added after Sigmoid / Softplus must be the column-index or the row-index grid");
            y.post_grid = (is_x && x.W > 1) || !is_y ? 1 : 2;
        } else
            fail(&n, "after Sigmoid / Softplus only `+ cell-index grid` followed by `* scalar` can be evaluated (the restore_coor form)");
        y.producer = -1;
        vals[n.out.at(0)] = y;
        return true;
    }

    void arithmetic(const o_node& n)
    {
        const val a = get(n, 0), b = get(n, 1);
        if (post_arithmetic(n, a, b, false) || post_arithmetic(n, b, a, true))
            return;
        if (a.kind == val::CONST && b.kind == val::CONST) { // unfolded exports: constants combined in the graph
            const o_tensor &ta = *a.c, &tb = *b.c;
            const size_t na = ta.count(), nb = tb.count();
            if (na != nb && na != 1 && nb != 1)
                fail(&n, "constant operands need equal sizes or a scalar");
            auto r = std::make_shared<o_tensor>();
            r->dtype = 1, r->dims = na >= nb ? ta.dims : tb.dims;
            const size_t cnt = std::max(na, nb);
            r->f.resize(cnt);
            for (size_t k = 0; k < cnt; ++k) {
                const float x = ta.f.at(na == 1 ? 0 : k), y = tb.f.at(nb == 1 ? 0 : k);
                r->f[k] = n.op == "Add" ? x + y : n.op == "Sub" ? x - y : n.op == "Mul" ? x * y : x / y;
            }
            val v;
            v.kind = val::CONST, v.c = r;
            vals[n.out.at(0)] = v;
            return;
        }
        if (a.kind == val::CONST || b.kind == val::CONST) {
            const bool const_first = a.kind == val::CONST;
            const val& x = const_first ? b : a;
            const std::vector<float> c = per_channel(n, const_first ? a : b, x.C);
            std::vector<float> s((size_t)x.C, 1.f), t((size_t)x.C, 0.f);
            for (int k = 0; k < x.C; ++k) {
                if (n.op == "Add")
                    t[k] = c[k];
                else if (n.op == "Mul")
                    s[k] = c[k];
                else if (n.op == "Sub") {
                    if (const_first)
                        s[k] = -1.f, t[k] = c[k];
                    else
                        t[k] = -c[k];
                } else { // Div
                    if (const_first)
                        fail(&n, "constant / map");
                    s[k] = 1.f / c[k];
                }
            }
            if (x.kind == val::IMAGE) {
                val y = x;
                if (x.pad[0] | x.pad[1] | x.pad[2] | x.pad[3])
                    fail(&n, "input normalisation after a Pad node (the padding would no longer be zero)");
                for (int k = 0; k < 3; ++k)
                    y.a[k] = x.a[k] * s[k], y.b[k] = x.b[k] * s[k] + t[k];
                vals[n.out.at(0)] = y;
                return;
            }
            need_map(n, x);
            val y = foldable(n.in[const_first ? 1 : 0], x) ? x : materialise(n, x);
            fold_affine(y.producer, s, t);
            vals[n.out.at(0)] = y;
            return;
        }
        if (n.op != "Add")
            fail(&n, "element-wise " + n.op + " of two feature maps");
        need_map(n, a), need_map(n, b);
        if (a.C != b.C || a.H != b.H || a.W != b.W)
            fail(&n, "operands differ in shape (broadcasting is not supported)");
        // residual form: the LATER convolution takes the other operand as its residual input
        auto can_take = [&](const std::string& name, const val& x, const val& other) {
            if (x.producer < 0 || other.coff != 0)
                return false;
            const hp_layer& L = m.layers[x.producer];
            if (L.op != HP_OP_CONV || L.res >= 0 || L.in == 0 || uses[name] != 1)
                return false;
            auto lw = last_writer.find(other.tensor);
            return lw != last_writer.end() && lw->second < x.producer;
        };
        val y;
        if (can_take(n.in[1], b, a)) {
            y = b;
            m.layers[y.producer].res = a.tensor;
        } else if (can_take(n.in[0], a, b)) {
            y = a;
            m.layers[y.producer].res = b.tensor;
        } else {
            const bool a_carries = b.coff == 0;
            if (!a_carries && a.coff != 0)
                fail(&n, "both operands live at a channel offset of a concatenation");
            y = materialise(n, a_carries ? a : b);
            m.layers[y.producer].res = a_carries ? b.tensor : a.tensor;
        }
        m.layers[y.producer].res_before_act = 0;
        vals[n.out.at(0)] = y;
    }

    // can the producer of v write its result at channel `off` of another tensor instead (no copy)?
    bool movable(const val& v, int off) const
    {
        const int old = v.tensor;
        auto tc = tensor_c.find(old);
        if (v.producer < 0 || v.coff != 0 || m.layers[v.producer].out_coff != 0 || tc == tensor_c.end() || tc->second != v.C || in_concat.count(old))
            return false;
        if (m.layers[v.producer].op != HP_OP_CONV && off % 8)
            return false; // depthwise / pool kernels store 8 channels at a time
        for (const hp_layer& L : m.layers) {
            if (L.in == old && (L.in_coff + off) % 8)
                return false; // readers of the moved map need 8-aligned channel offsets
            if (L.res == old && off != 0)
                return false; // residual inputs are read from channel 0
            if (L.out == old && &L != &m.layers[v.producer])
                return false;
        }
        return true;
    }
    void move_into(const val& v, int T, int off)
    {
        const int old = v.tensor;
        for (hp_layer& L : m.layers) {
            if (L.in == old)
                L.in = T, L.in_coff += off;
            if (L.res == old)
                L.res = T;
        }
        hp_layer& P = m.layers[v.producer];
        P.out = T, P.out_coff = off;
        for (auto& kv : vals)
            if (kv.second.kind == val::MAP && kv.second.tensor == old)
                kv.second.tensor = T, kv.second.coff += off;
        last_writer[T] = std::max(last_writer.count(T) ? last_writer[T] : -1, v.producer);
        last_writer.erase(old), tensor_c.erase(old);
    }

    // Multi-stage heads (OpenPose: every stage reads concat(stage outputs, backbone features)): when one input already sits
    // in an earlier concatenation at the offset it would get here, and what the other slots of that tensor hold is dead by the
    // time their new contents are produced, the new members are written over the old ones - the features are never copied.
    bool concat_in_place(const o_node& n, const std::vector<val>& in, const std::vector<int>& offs, int total, val& y)
    {
        for (size_t j = 0; j < in.size(); ++j) {
            const int T = in[j].tensor;
            if (!in_concat.count(T) || in[j].coff != offs[j] || tensor_c[T] != total)
                continue;
            bool ok = true;
            for (size_t k = 0; k < in.size() && ok; ++k) {
                if (in[k].tensor == T && in[k].coff == offs[k])
                    continue; // already in place
                if (!movable(in[k], offs[k])) {
                    ok = false;
                    break;
                }
                const int lo = offs[k], hi = offs[k] + in[k].C, pk = in[k].producer;
                for (size_t i = 0; i < m.layers.size() && ok; ++i) {
                    const hp_layer& L = m.layers[i];
                    const bool reads = (L.in == T && L.in_coff < hi && L.in_coff + L.cin > lo) || (L.res == T && L.cout > lo);
                    const bool writes = L.out == T && L.out_coff < hi && L.out_coff + L.cout > lo;
                    if ((reads || writes) && (int)i >= pk)
                        ok = false; // somebody still touches the old contents after the new ones are written
                }
                for (const auto& kv : vals) {
                    const val& o = kv.second;
                    if (o.kind == val::MAP && o.tensor == T && o.coff < hi && o.coff + o.C > lo && remaining[kv.first] > 0)
                        ok = false; // a node further down (or a graph output) still wants the old contents
                }
            }
            if (!ok)
                continue;
            for (size_t k = 0; k < in.size(); ++k) {
                if (in[k].tensor == T && in[k].coff == offs[k])
                    continue;
                const int lo = offs[k], hi = offs[k] + in[k].C;
                for (auto& kv : vals)
                    if (kv.second.kind == val::MAP && kv.second.tensor == T && kv.second.coff < hi && kv.second.coff + kv.second.C > lo)
                        kv.second.kind = val::NONE; // overwritten from here on
                move_into(in[k], T, offs[k]);
            }
            y.kind = val::MAP, y.tensor = T, y.coff = 0, y.C = total, y.H = in[j].H, y.W = in[j].W, y.producer = -1;
            (void)n;
            return true;
        }
        return false;
    }

    void concat(const o_node& n)
    {
        if (n.geti("axis", 1) != 1)
            fail(&n, "only channel concatenation (axis = 1) is supported");
        std::vector<val> in;
        std::vector<int> offs;
        int total = 0;
        for (size_t k = 0; k < n.in.size(); ++k) {
            const val v = get(n, k);
            need_map(n, v);
            if (v.H != get(n, 0).H || v.W != get(n, 0).W)
                fail(&n, "inputs differ in size");
            in.push_back(v), offs.push_back(total);
            total += v.C;
        }
        val y;
        if (concat_in_place(n, in, offs, total, y)) {
            vals[n.out.at(0)] = y;
            return;
        }
        const int T = m.new_tensor();
        for (size_t k = 0; k < in.size(); ++k) {
            const val v = get(n, k); // (re-read: an earlier member of this very concat may have moved it)
            if (movable(v, offs[k]))
                move_into(v, T, offs[k]);
            else
                identity(n, v, T, offs[k]);
        }
        tensor_c[T] = total;
        in_concat.insert(T);
        y.kind = val::MAP, y.tensor = T, y.C = total, y.H = in[0].H, y.W = in[0].W;
        vals[n.out.at(0)] = y;
    }

    // shape-only operators on constants (PReLU slopes, normalisation constants in exports without constant folding)
    void reshape_constant(const o_node& n)
    {
        const o_tensor& t = *get(n, 0).c;
        auto r = std::make_shared<o_tensor>(t);
        std::vector<int64_t> arg;
        if (const o_attr* a = n.attr("axes"))
            arg = a->ints;
        else if (has_in(n, 1)) {
            const val& c = get(n, 1);
            if (c.kind != val::CONST)
                fail(&n, "axes / shape must be constant");
            arg = c.c->i;
        }
        if (n.op == "Unsqueeze") {
            const int64_t nd = (int64_t)t.dims.size() + (int64_t)arg.size();
            for (auto& ax : arg)
                ax = ax < 0 ? ax + nd : ax;
            std::sort(arg.begin(), arg.end());
            for (int64_t ax : arg) {
                if (ax < 0 || ax > (int64_t)r->dims.size())
                    fail(&n, "axis out of range");
                r->dims.insert(r->dims.begin() + ax, 1);
            }
        } else if (n.op == "Squeeze") {
            std::vector<int64_t> d;
            for (size_t k = 0; k < t.dims.size(); ++k) {
                bool drop = arg.empty() ? t.dims[k] == 1 : false;
                for (int64_t ax : arg)
                    if ((ax < 0 ? ax + (int64_t)t.dims.size() : ax) == (int64_t)k)
                        drop = true;
                if (!drop)
                    d.push_back(t.dims[k]);
            }
            r->dims = d;
        } else if (n.op == "Reshape") {
            int64_t known = 1, infer = -1;
            for (size_t k = 0; k < arg.size(); ++k) {
                if (arg[k] == 0)
                    arg[k] = k < t.dims.size() ? t.dims[k] : 1;
                if (arg[k] == -1)
                    infer = (int64_t)k;
                else
                    known *= arg[k];
            }
            if (infer >= 0)
                arg[infer] = known ? (int64_t)t.count() / known : 0;
            r->dims = arg;
            if (r->count() != t.count())
                fail(&n, "shape does not match the constant");
        }
        val v;
        v.kind = val::CONST, v.c = r;
        vals[n.out.at(0)] = v;
    }

    void pad(const o_node& n)
    {
        val x = get(n, 0);
        std::vector<int64_t> p;
        if (const o_attr* a = n.attr("pads"))
            p = a->ints;
        else if (has_in(n, 1)) {
            const val& c = get(n, 1);
            if (c.kind != val::CONST)
                fail(&n, "pads must be constant");
            p = c.c->i;
        }
        const o_attr* mode = n.attr("mode");
        if (mode && mode->s != "constant")
            fail(&n, "mode '" + mode->s + "'");
        float value = n.getf("value", 0.f);
        if (has_in(n, 2)) {
            const val& c = get(n, 2);
            if (c.kind == val::CONST && c.c->count() == 1)
                value = c.c->f[0];
        }
        if (p.size() != 8 || p[0] || p[1] || p[4] || p[5] || value != 0.f)
            fail(&n, "only zero padding of H and W of a 4-D map is supported");
        if (x.kind != val::MAP && !(x.kind == val::IMAGE && !x.nhwc))
            fail(&n, "expects a feature map");
        x.pad[0] += (int)p[2], x.pad[1] += (int)p[3], x.pad[2] += (int)p[6], x.pad[3] += (int)p[7];
        x.producer = -1; // nothing may be folded through the padding
        vals[n.out.at(0)] = x;
    }

    void run(int in_w, int in_h)
    {
        // ---- the one network input (src/tensorrt.cpp:179-190)
        std::vector<const o_value_info*> real_inputs;
        for (const auto& vi : g.inputs)
            if (!g.init.count(vi.name))
                real_inputs.push_back(&vi);
        if (real_inputs.size() != 1)
            fail(nullptr, "detected " + std::to_string(real_inputs.size()) + " inputs (only one-input models are supported)");
        const o_value_info& in = *real_inputs[0];
        val img;
        img.kind = val::IMAGE, img.C = 3;
        int64_t gh = -1, gw = -1;
        if (in.has_shape) {
            const auto& d = in.dims;
            if (d.size() != 3 && d.size() != 4)
                fail(nullptr, "input '" + in.name + "' must have 3 or 4 dimensions");
            const size_t o = d.size() - 3;
            if (d[o] == 3)
                gh = d[o + 1], gw = d[o + 2];
            else if (d.size() == 4 && d[3] == 3)
                gh = d[1], gw = d[2], img.nhwc = true;
            else
                fail(nullptr, "input '" + in.name + "': the channel dimension must be 3");
        }
        if (in_w > 0 && in_h > 0) {
            if ((gh > 0 && gh != in_h) || (gw > 0 && gw != in_w))
                fail(nullptr, "the model is fixed to " + std::to_string(gw) + "x" + std::to_string(gh) + " (WxH), asked for " + std::to_string(in_w) + "x" + std::to_string(in_h));
        } else {
            if (gh <= 0 || gw <= 0)
                fail(nullptr, "the model's input size is dynamic: pass the input size");
            in_w = (int)gw, in_h = (int)gh;
        }
        img.H = in_h, img.W = in_w;
        m.in_w = in_w, m.in_h = in_h;
        vals[in.name] = img;

        for (const auto& n : g.nodes)
            for (const auto& s : n.in)
                ++uses[s];
        for (const auto& o : g.outputs)
            ++uses[o.name];
        remaining = uses;

        for (const auto& n : g.nodes) {
            if (n.out.empty())
                continue;
            const std::string& op = n.op;
            if (op == "Constant") {
                const o_attr* a = n.attr("value");
                if (!a || !a->t)
                    fail(&n, "only tensor-valued constants are supported");
                val v;
                v.kind = val::CONST, v.c = a->t;
                vals[n.out[0]] = v;
            } else if (op == "Conv")
                conv(n);
            else if (op == "MaxPool")
                maxpool(n);
            else if (op == "Resize" || op == "Upsample")
                resize(n);
            else if (op == "BatchNormalization")
                batchnorm(n);
            else if (op == "Relu")
                activation(n, HP_ACT_RELU, 0.f, nullptr);
            else if (op == "LeakyRelu")
                activation(n, HP_ACT_LEAKY, n.getf("alpha", 0.01f), nullptr);
            else if (op == "PRelu") {
                const val& x = get(n, 0);
                const val& s = get(n, 1);
                if (s.kind != val::CONST)
                    fail(&n, "slope must be an initializer");
                const std::vector<float> alpha = per_channel(n, s, x.C);
                activation(n, HP_ACT_PRELU, 0.f, &alpha);
            } else if (op == "Clip") {
                float lo = n.getf("min", -INFINITY), hi = n.getf("max", INFINITY);
                if (has_in(n, 1)) {
                    const val& c = get(n, 1);
                    if (c.kind != val::CONST || c.c->count() != 1)
                        fail(&n, "min must be a scalar constant");
                    lo = c.c->f[0];
                }
                if (has_in(n, 2)) {
                    const val& c = get(n, 2);
                    if (c.kind != val::CONST || c.c->count() != 1)
                        fail(&n, "max must be a scalar constant");
                    hi = c.c->f[0];
                }
                if (lo == 0.f && hi == 6.f)
                    activation(n, HP_ACT_RELU6, 0.f, nullptr);
                else if (lo == 0.f && std::isinf(hi))
                    activation(n, HP_ACT_RELU, 0.f, nullptr);
                else
                    fail(&n, "only Clip(0, 6) and Clip(0, inf) are supported");
            } else if (op == "Sigmoid" || op == "Softplus") {
                val x = get(n, 0);
                const bool nhwc = x.kind == val::MAP && x.nhwc_view && x.view.empty(); // (element-wise: see activation())
                x.nhwc_view = nhwc ? false : x.nhwc_view;
                need_map(n, x);
                x.post_act = op == "Sigmoid" ? HP_ACT_SIGMOID : HP_ACT_SOFTPLUS;
                x.producer = -1;
                x.nhwc_view = nhwc;
                vals[n.out[0]] = x;
            } else if (op == "Add" || op == "Sub" || op == "Mul" || op == "Div")
                arithmetic(n);
            else if (op == "Concat")
                concat(n);
            else if (op == "Pad")
                pad(n);
            else if (op == "Identity" || op == "Dropout")
                vals[n.out[0]] = get(n, 0);
            else if ((op == "Unsqueeze" || op == "Squeeze" || op == "Reshape" || op == "Cast") && get(n, 0).kind == val::CONST)
                reshape_constant(n);
            else if (op == "Reshape" || op == "Flatten" || ((op == "Squeeze" || op == "Unsqueeze") && get(n, 0).kind == val::MAP)) {
                // post-processing of a head (e.g. the PoseProposal edge tensor [17*9*9, 12, 12] -> [17, 9, 9, 12, 12]): the memory of a
                // contiguous NCHW map does not change, and the parsers take shapes from their own arguments - a view that may only go on
                // to a graph output (or another Reshape)
                val x = get(n, 0);
                if (x.kind != val::MAP || x.nhwc_view)
                    fail(&n, "expects a feature map in N,C,H,W order");
                if (x.pad[0] | x.pad[1] | x.pad[2] | x.pad[3])
                    fail(&n, "cannot reshape a padded map");
                std::vector<int64_t> shape;
                if (op == "Reshape") {
                    if (!has_in(n, 1) || get(n, 1).kind != val::CONST)
                        fail(&n, "the target shape must be a constant");
                    shape = get(n, 1).c->i;
                    const int64_t per_frame = (int64_t)x.C * x.H * x.W;
                    int64_t known = 1, infer = -1;
                    for (size_t k = 0; k < shape.size(); ++k) {
                        if (shape[k] == 0) // "copy from the input": only meaningful on the batch axis here
                            shape[k] = k == 0 ? 1 : -2;
                        if (shape[k] == -1)
                            infer = (int64_t)k;
                        else if (shape[k] > 0)
                            known *= shape[k];
                        else
                            fail(&n, "unsupported target shape");
                    }
                    // the batch axis may be written as -1, 0 or 1; everything after it must account for exactly one frame
                    if (infer >= 0 && infer != 0) {
                        if (known <= 0 || per_frame % known)
                            fail(&n, "target shape does not divide the feature map");
                        shape[infer] = per_frame / known, known = per_frame;
                    } else if (infer == 0)
                        shape[0] = 1;
                    int64_t total = 1;
                    for (int64_t d : shape)
                        total *= d;
                    if (total != per_frame)
                        fail(&n, "target shape holds " + std::to_string(total) + " elements per frame, the feature map " + std::to_string(per_frame));
                } else
                    shape = { 1, (int64_t)x.C * x.H * x.W };
                x.view = shape;
                x.producer = -1;
                vals[n.out[0]] = x;
            } else if (op == "Split" || op == "Slice") {
                // channel ranges of a head's output (PoseProposal: pc, pi, px, py, pw, ph | pe): zero-copy views at a channel offset
                val x = get(n, 0);
                if (x.kind != val::MAP || !x.view.empty() || x.nhwc_view) // (a pending Sigmoid / Softplus travels with every slice)
                    fail(&n, "expects a feature map in N,C,H,W order");
                if (x.pad[0] | x.pad[1] | x.pad[2] | x.pad[3])
                    fail(&n, "cannot slice a padded map");
                auto ints_of = [&](const char* attr, size_t input) -> std::vector<int64_t> {
                    if (const o_attr* a = n.attr(attr))
                        return a->ints;
                    if (has_in(n, input)) {
                        const val& c = get(n, input);
                        if (c.kind != val::CONST)
                            fail(&n, std::string(attr) + " must be constant");
                        return c.c->i;
                    }
                    return {};
                };
                if (op == "Split") {
                    int64_t axis = n.geti("axis", 0);
                    if (axis < 0)
                        axis += 4;
                    if (axis != 1)
                        fail(&n, "only a split along the channel axis is supported");
                    std::vector<int64_t> sizes = ints_of("split", 1);
                    if (sizes.empty()) {
                        if (n.out.empty() || x.C % (int)n.out.size())
                            fail(&n, "channels do not divide evenly over the outputs");
                        sizes.assign(n.out.size(), x.C / (int64_t)n.out.size());
                    }
                    if (sizes.size() != n.out.size())
                        fail(&n, "split sizes do not match the outputs");
                    int64_t at = 0;
                    for (size_t k = 0; k < sizes.size(); ++k) {
                        if (sizes[k] <= 0 || at + sizes[k] > x.C)
                            fail(&n, "split sizes exceed the channel count");
                        val y = x;
                        y.coff = x.coff + (int)at, y.C = (int)sizes[k], y.producer = -1;
                        vals[n.out[k]] = y;
                        at += sizes[k];
                    }
                } else {
                    const std::vector<int64_t> starts = ints_of("starts", 1), ends = ints_of("ends", 2);
                    std::vector<int64_t> axes = ints_of("axes", 3), steps = ints_of("steps", 4);
                    if (starts.size() != 1 || ends.size() != 1 || (axes.size() > 1) || (steps.size() > 1) || (!steps.empty() && steps[0] != 1))
                        fail(&n, "only a unit-step slice of one axis is supported");
                    int64_t axis = axes.empty() ? 0 : axes[0];
                    if (axis < 0)
                        axis += 4;
                    if (axis != 1)
                        fail(&n, "only a slice along the channel axis is supported");
                    int64_t s0 = starts[0] < 0 ? starts[0] + x.C : starts[0], e0 = ends[0] < 0 ? ends[0] + x.C : ends[0];
                    s0 = std::max<int64_t>(0, std::min<int64_t>(s0, x.C)), e0 = std::max<int64_t>(0, std::min<int64_t>(e0, x.C));
                    if (e0 <= s0)
                        fail(&n, "empty slice");
                    val y = x;
                    y.coff = x.coff + (int)s0, y.C = (int)(e0 - s0), y.producer = -1;
                    vals[n.out[0]] = y;
                }
            } else if (op == "Transpose") {
                val x = get(n, 0);
                const o_attr* perm = n.attr("perm");
                const std::vector<int64_t> pv = perm ? perm->ints : std::vector<int64_t>{};
                const bool to_nchw = pv == std::vector<int64_t>{ 0, 3, 1, 2 }, to_nhwc = pv == std::vector<int64_t>{ 0, 2, 3, 1 };
                const bool identity = pv == std::vector<int64_t>{ 0, 1, 2, 3 };
                if (x.kind == val::IMAGE) {
                    if (!x.nhwc || !to_nchw)
                        fail(&n, "only the N,H,W,3 -> N,3,H,W transpose of the input is supported");
                    x.nhwc = false;
                } else if (x.kind == val::MAP && x.view.empty()) {
                    // exporters that keep TensorFlow's layout wrap operators in N,C,H,W <-> N,H,W,C pairs: a pair cancels; a map left in
                    // N,H,W,C cannot be handed to the parsers (they index [C,H,W], include/hyperpose/utility/data.hpp:22-23)
                    if (identity)
                        ;
                    else if (!x.nhwc_view && to_nhwc)
                        x.nhwc_view = true;
                    else if (x.nhwc_view && to_nchw)
                        x.nhwc_view = false;
                    else
                        fail(&n, "unsupported permutation of a feature map (N,C,H,W <-> N,H,W,C pairs and the identity are supported)");
                    if (uses[n.in[0]] != 1)
                        x.producer = -1; // (somebody else reads the map as it was: an activation behind the Transpose must not be folded into its producer)
                } else
                    fail(&n, "cannot transpose this value");
                vals[n.out[0]] = x;
            } else
                fail(&n, "operator not supported by the importer");
            for (const auto& s : n.in)
                --remaining[s];
        }

        // ---- outputs
        for (const auto& o : g.outputs) {
            auto it = vals.find(o.name);
            if (it == vals.end() || it->second.kind != val::MAP)
                fail(nullptr, "graph output '" + o.name + "' is not a feature map produced by the graph");
            const val& v = it->second;
            if (v.pad[0] | v.pad[1] | v.pad[2] | v.pad[3])
                fail(nullptr, "graph output '" + o.name + "' is a padded map");
            if (v.nhwc_view)
                fail(nullptr, "graph output '" + o.name + "' is left in N,H,W,C order: the parsers index [C,H,W]");
            m.output(o.name.c_str(), v.tensor, v.coff, v.C, v.post_act);
            // (hp_output_desc::scale = 0 means "no factor": a graph that really multiplies its output by 0 - or by a non-finite constant -
            // cannot be expressed and is refused rather than silently losing the factor)
            if (v.post_scale == 0.f || !std::isfinite(v.post_scale))
                fail(nullptr, "graph output '" + o.name + "' is scaled by " + std::to_string(v.post_scale) + ": not representable as an output post-op");
            m.outputs.back().grid = v.post_grid, m.outputs.back().scale = v.post_scale == 1.f ? 0.f : v.post_scale; // (0 = no factor)
        }
        if (m.layers.empty())
            fail(nullptr, "the graph has no convolution");

        // ---- compact tensor numbering (moved concat members leave holes)
        std::map<int, int> remap;
        remap[0] = 0;
        auto id = [&](int t) {
            if (t < 0)
                return t;
            auto it = remap.find(t);
            if (it == remap.end())
                it = remap.emplace(t, (int)remap.size()).first;
            return it->second;
        };
        for (hp_layer& L : m.layers) {
            L.in = id(L.in);
            L.res = id(L.res);
            L.out = id(L.out);
        }
        for (hp_output_desc& o : m.outputs)
            o.tensor = id(o.tensor);
        m.next_tensor = (int)remap.size();
        m.n_weights = (int64_t)blob.size();
    }
};

int import_bytes(hp_model** out, const uint8_t* data, size_t size, int in_w, int in_h)
{
    auto m = std::make_unique<hp_model>();
    try {
        o_graph g;
        std::string why;
        if (!parse_model(data, size, g, why))
            throw lower_error{ "onnx: " + printable(why) };
        m->arch = "onnx:" + printable(g.name);
        lowering lw(g, *m);
        lw.run(in_w, in_h);
    } catch (const lower_error& e) {
        HP_REQUIRE(false, HP_ERR_INVALID, "%s", e.msg.c_str());
    } catch (const std::out_of_range&) {
        HP_REQUIRE(false, HP_ERR_INVALID, "onnx: a tensor holds fewer values than its shape says");
    } catch (const std::exception& e) { // bad_alloc / length_error from absurd sizes in a corrupted file
        HP_REQUIRE(false, HP_ERR_INVALID, "onnx: malformed model (%s)", e.what());
    }
    *out = m.release();
    return HP_OK;
}

} // namespace

extern "C" {

int hp_model_from_onnx(hp_model** out, const void* data, size_t size, int in_w, int in_h)
{
    HP_REQUIRE(out && data && size > 0, HP_ERR_INVALID, "hp_model_from_onnx: null argument");
    return import_bytes(out, (const uint8_t*)data, size, in_w, in_h);
}

int hp_model_from_onnx_file(hp_model** out, const char* path, int in_w, int in_h)
{
    HP_REQUIRE(out && path, HP_ERR_INVALID, "hp_model_from_onnx_file: null argument");
    FILE* f = fopen(path, "rb");
    HP_REQUIRE(f, HP_ERR_INVALID, "hp_model_from_onnx_file: cannot open %s", path);
    std::vector<uint8_t> bytes;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0)
        bytes.insert(bytes.end(), buf, buf + got);
    fclose(f);
    HP_REQUIRE(!bytes.empty(), HP_ERR_INVALID, "hp_model_from_onnx_file: %s is empty", path);
    return import_bytes(out, bytes.data(), bytes.size(), in_w, in_h);
}

int hp_model_weights(const hp_model* m, const float** blob, size_t* n)
{
    HP_REQUIRE(m && blob && n, HP_ERR_INVALID, "hp_model_weights: null argument");
    *blob = m->weights.empty() ? nullptr : m->weights.data();
    *n = m->weights.size();
    return HP_OK;
}

int hp_model_input_size(const hp_model* m, int* w, int* h)
{
    HP_REQUIRE(m && w && h, HP_ERR_INVALID, "hp_model_input_size: null argument");
    *w = m->in_w, *h = m->in_h;
    return HP_OK;
}

} // extern "C"
