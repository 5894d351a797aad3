// Canonical form of a folded graph, so that the lowering (lower.cpp) sees ONE spelling of each construct whatever the exporter,
// the opset or an optimiser pass wrote.  The reference hands any ONNX file to TensorRT's parser (img2img_build.cpp:81-88), which
// does this normalisation in its importer; tests/test_loader_rewrites.py generates the spellings below from each exported graph
// and checks that they lower to the plan of the original.
//
//   * Identity, Dropout (inference), float -> float Cast, Transpose by the identity permutation, and Reshape / Flatten / Squeeze /
//     Unsqueeze that keep the shape: removed, their output becomes a name for their input.
//   * Transpose of a Transpose: one Transpose by the composed permutation (or nothing when that is the identity).
//   * Flatten / Squeeze / Unsqueeze on runtime tensors: Reshape to the inferred shape; a Reshape of a Reshape reads the first
//     one's input.
//   * Gemm with a constant B (and C): MatMul by alpha * op(B) + Add of beta * C.
//   * The 2-D sandwich exporters put around a product of higher-rank rows - Reshape([-1, K]) -> MatMul -> [Add] -> Reshape(lead.., N):
//     the product on the rows where they are (the two Reshapes go, the shapes in between get their leading dimensions back).
//   * Node order: depth-first from the graph output, operands in slot order - a function of the graph alone, not of the order the
//     file lists its nodes in; nodes the output does not depend on are dropped.
#include <algorithm>
#include <functional>
#include <set>
#include <stdexcept>

#include "fold.h"

namespace w2x {

namespace {

using Shape = std::vector<int64_t>;

bool is_reshape_like(const std::string& op) { return op == "Reshape" || op == "Flatten" || op == "Squeeze" || op == "Unsqueeze"; }
bool float_dtype(int dt) { return dt == DT_F32 || dt == DT_F16 || dt == DT_F64; }

struct Simplifier {
    FoldedGraph& g;
    std::vector<Node*> nodes;                       // working list, topological
    std::map<std::string, Node*> producer;
    std::map<std::string, std::vector<Node*>> consumers;
    int fresh = 0;

    explicit Simplifier(FoldedGraph& g_) : g(g_) {}

    void index() {
        producer.clear(); consumers.clear();
        for (Node* n : nodes) {
            for (auto& o : n->out) if (!o.empty()) producer[o] = n;
            for (auto& i : n->in) if (!i.empty() && !g.is_const(i)) consumers[i].push_back(n);
        }
    }
    Node* only_consumer(const std::string& name) {
        if (name == g.output) return nullptr;
        auto it = consumers.find(name);
        return it != consumers.end() && it->second.size() == 1 ? it->second[0] : nullptr;
    }
    const Shape& shape(const std::string& n) { return g.val(n).shape; }

    // every use of `from` reads `to` from now on
    void alias(const std::string& from, const std::string& to) {
        for (Node* n : nodes) for (auto& i : n->in) if (i == from) i = to;
        if (g.output == from) {
            // the graph output keeps its name: the producer of `to` writes it instead (the input itself cannot be the output)
            auto it = producer.find(to);
            if (it == producer.end()) throw std::runtime_error("graph: the output is the input");
            for (auto& o : it->second->out) if (o == to) o = from;
            for (Node* n : nodes) for (auto& i : n->in) if (i == to) i = from;
            g.vals[from] = g.vals[to];
        }
    }
    void erase(Node* n) { nodes.erase(std::find(nodes.begin(), nodes.end(), n)); }

    std::string new_const(HTensorP t) {
        std::string name = "__w2x_c" + std::to_string(fresh++);
        Value v; v.is_const = true; v.c = t; v.shape = t->shape; v.dtype = t->dtype;
        g.vals[name] = v;
        return name;
    }
    std::string shape_const(const Shape& s) {
        auto t = std::make_shared<HTensor>(); t->dtype = DT_I64; t->shape = {(int64_t)s.size()}; t->i = s;
        return new_const(t);
    }
    Node* new_node(const Node& like, size_t after_index) {
        g.owned.push_back(like);
        Node* n = &g.owned.back();
        nodes.insert(nodes.begin() + (long)after_index + 1, n);
        return n;
    }

    // one rewrite; true if the graph changed.  products_only: the Gemm and 2-D sandwich rules alone - they run to their fixed point first, so that the
    // Reshape-of-a-Reshape rule does not merge a sandwich's Reshapes into their neighbours (a window partition's last Reshape in front, the q / k / v
    // split behind) before the sandwich is recognised
    bool step(bool products_only) {
        index();
        for (size_t idx = 0; idx < nodes.size(); ++idx) {
            Node* n = nodes[idx];
            const std::string& op = n->op;
            if (n->in.empty() || n->in[0].empty() || g.is_const(n->in[0])) continue;
            const std::string x = n->in[0];
            if (products_only && op != "Gemm" && op != "Reshape") continue;
            if (products_only && op == "Reshape") { if (sandwich(n, x)) return true; continue; }
            // ---- no-ops
            bool noop = op == "Identity" || op == "Dropout";
            if (op == "Cast") noop = float_dtype(g.val(x).dtype) && float_dtype((int)n->ai("to", DT_F32));
            if (is_reshape_like(op)) noop = shape(n->out[0]) == shape(x);
            if (op == "Transpose") {
                auto perm = n->aints("perm");
                noop = !perm.empty();
                for (size_t k = 0; k < perm.size(); ++k) noop = noop && perm[k] == (int64_t)k;
            }
            if (noop && !(x == g.input && n->out[0] == g.output)) { alias(n->out[0], x); erase(n); return true; }
            // ---- Transpose of a Transpose
            if (op == "Transpose") {
                auto it = producer.find(x);
                if (it != producer.end() && it->second->op == "Transpose" && only_consumer(x) == n) {
                    Node* t1 = it->second;
                    auto p1 = t1->aints("perm"), p2 = n->aints("perm");
                    const size_t r = shape(x).size();
                    if (p1.empty()) for (size_t k = 0; k < r; ++k) p1.push_back((int64_t)(r - 1 - k));
                    if (p2.empty()) for (size_t k = 0; k < r; ++k) p2.push_back((int64_t)(r - 1 - k));
                    std::vector<int64_t> pc(r);
                    for (size_t k = 0; k < r; ++k) pc[k] = p1[(size_t)p2[k]];
                    n->in[0] = t1->in[0];
                    Attr a; a.type = 7; a.ints = pc; n->attr["perm"] = a;
                    erase(t1);
                    return true;
                }
            }
            // ---- Flatten / Squeeze / Unsqueeze spelled as Reshape; Reshape of a Reshape
            if (op == "Flatten" || op == "Squeeze" || op == "Unsqueeze") {
                n->op = "Reshape"; n->attr.clear();
                n->in = {x, shape_const(shape(n->out[0]))};
                return true;
            }
            if (op == "Reshape") {
                auto it = producer.find(x);
                if (it != producer.end() && it->second->op == "Reshape" && only_consumer(x) == n) {
                    Node* r1 = it->second;
                    n->in = {r1->in[0], shape_const(shape(n->out[0]))};
                    n->attr.clear();
                    erase(r1);
                    return true;
                }
            }
            // ---- Gemm with constant B (and C) -> MatMul [+ Add]
            if (op == "Gemm" && n->in.size() >= 2 && g.is_const(n->in[1]) && n->ai("transA", 0) == 0 && (n->in.size() < 3 || n->in[2].empty() || g.is_const(n->in[2]))) {
                const HTensor& b = g.cst(n->in[1]);
                if (b.rank() != 2 || !b.is_float()) throw std::runtime_error("graph: Gemm \"" + n->name + "\" with a weight that is not a 2-D float matrix");
                const float alpha = n->af("alpha", 1.f), beta = n->af("beta", 1.f);
                const bool tb = n->ai("transB", 0) != 0;
                const int64_t K = tb ? b.shape[1] : b.shape[0], N = tb ? b.shape[0] : b.shape[1];
                auto w = std::make_shared<HTensor>(); w->dtype = DT_F32; w->shape = {K, N}; w->f.resize((size_t)(K * N));
                for (int64_t k = 0; k < K; ++k) for (int64_t c = 0; c < N; ++c) w->f[(size_t)(k * N + c)] = alpha * (tb ? b.f[(size_t)(c * K + k)] : b.f[(size_t)(k * N + c)]);
                const std::string out = n->out[0];
                std::string bias;
                if (n->in.size() > 2 && !n->in[2].empty()) {
                    const HTensor& c = g.cst(n->in[2]);
                    if (!c.is_float() || (c.numel() != N && c.numel() != 1)) throw std::runtime_error("graph: Gemm \"" + n->name + "\" with a bias that is not a row of N values");
                    auto bb = std::make_shared<HTensor>(); bb->dtype = DT_F32; bb->shape = {N}; bb->f.resize((size_t)N);
                    for (int64_t c2 = 0; c2 < N; ++c2) bb->f[(size_t)c2] = beta * c.f[c.numel() == 1 ? 0 : (size_t)c2];
                    bias = new_const(bb);
                }
                n->op = "MatMul"; n->attr.clear();
                n->in = {x, new_const(w)};
                if (!bias.empty()) {
                    const std::string mid = "__w2x_t" + std::to_string(fresh++);
                    g.vals[mid] = g.vals[out];
                    n->out[0] = mid;
                    Node add; add.op = "Add"; add.name = n->name + "/bias"; add.in = {mid, bias}; add.out = {out};
                    new_node(add, idx);
                }
                return true;
            }
        }
        return false;
    }

    // The 2-D sandwich around a product of rows: n = Reshape(x -> [M, K]) (x of higher rank, last dimension kept) -> MatMul(const [K, N]) -> [Add(const)]
    // -> Reshape(s) only.  The product runs on the rows where they are: its result gets x's leading dimensions, the Reshape(s) behind it re-shape the
    // same elements in the same order (one that restores [lead.., N] is then a no-op and goes in the second phase).
    bool sandwich(Node* n, const std::string& x) {
        if (!(shape(n->out[0]).size() == 2 && shape(x).size() > 2 && shape(x).back() == shape(n->out[0])[1])) return false;
        Node* mm = only_consumer(n->out[0]);
        if (!(mm && mm->op == "MatMul" && mm->in[0] == n->out[0] && g.is_const(mm->in[1]) && g.cst(mm->in[1]).rank() == 2)) return false;
        Node* last = mm;
        Node* add = only_consumer(mm->out[0]);
        if (add && add->op == "Add" && ((add->in[0] == mm->out[0] && g.is_const(add->in[1])) || (add->in[1] == mm->out[0] && g.is_const(add->in[0])))) {
            // the Add joins the sandwich only when its constant means the same on [M, N] and on [lead.., N]: a scalar, or one value per column
            // ([N] / [1, N]).  An [M, N] or [M, 1] table is legal in the 2-D form and broadcasts differently (or not at all) once the rows
            // have their leading dimensions back: the sandwich then stays as the file wrote it.
            const HTensor& c = g.cst(add->in[0] == mm->out[0] ? add->in[1] : add->in[0]);
            const int64_t N = g.cst(mm->in[1]).shape[1];
            bool lead1 = true;
            for (size_t k = 0; k + 1 < c.shape.size(); ++k) lead1 = lead1 && c.shape[k] == 1;
            const bool per_column = c.numel() == 1 || (!c.shape.empty() && c.shape.back() == N && lead1);
            if (!per_column) return false;
            last = add;
        } else add = nullptr;
        if (last->out[0] == g.output) return false;
        auto us = consumers.find(last->out[0]);
        if (us == consumers.end() || us->second.empty()) return false;
        for (Node* u : us->second) if (u->op != "Reshape" || u->in[0] != last->out[0]) return false;
        Shape want(shape(x).begin(), shape(x).end() - 1);
        want.push_back(g.cst(mm->in[1]).shape[1]);
        mm->in[0] = x;
        g.vals[mm->out[0]].shape = want;
        if (add) g.vals[add->out[0]].shape = want;
        erase(n);
        return true;
    }

    void canonical_order() {
        index();
        std::vector<Node*> order;
        std::set<Node*> seen;
        // iterative depth-first walk from the output: a node is emitted after all its operands (slot order)
        std::vector<std::pair<Node*, size_t>> stack;
        auto push = [&](const std::string& name) {
            auto it = producer.find(name);
            if (it != producer.end() && !seen.count(it->second)) { seen.insert(it->second); stack.push_back({it->second, 0}); }
        };
        push(g.output);
        while (!stack.empty()) {
            auto& [n, k] = stack.back();
            if (k < n->in.size()) { const std::string in = n->in[k++]; if (!in.empty() && !g.is_const(in)) push(in); }
            else { order.push_back(n); stack.pop_back(); }
        }
        nodes.swap(order);
    }
};

}  // namespace

void simplify_graph(FoldedGraph& g) {
    Simplifier s(g);
    for (const Node* n : g.nodes) { g.owned.push_back(*n); s.nodes.push_back(&g.owned.back()); }
    int guard = 0;
    for (bool changed = true; changed;) {          // (a sandwich may only show once a no-op inside it is gone: both phases until neither finds anything)
        changed = false;
        for (int phase = 0; phase < 2; ++phase)
            while (s.step(phase == 0)) {
                changed = true;
                if (++guard > 1000000) throw std::runtime_error("graph: simplification does not terminate");
            }
    }
    s.canonical_order();
    g.nodes.assign(s.nodes.begin(), s.nodes.end());
    g.consumers.clear(); g.producer.clear();
    for (const Node* n : g.nodes) {
        for (auto& o : n->out) g.producer[o] = n;
        for (auto& i : n->in) if (!i.empty() && !g.is_const(i)) g.consumers[i].push_back(n);
    }
}

}  // namespace w2x
