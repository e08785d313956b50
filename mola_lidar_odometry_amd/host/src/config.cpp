// config.cpp -- YAML-subset reader, ${ENV|default} substitution and the run-time formula evaluator used for
// the expressions of pipelines/lidar3d-default.yaml:190,198 (mp2p_icp::Parameterizable [U] on top of
// mrpt::expr [U]).  Host-side, tiny, not on the hot path.
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "mp2p_icp_hip/mp2p_icp_hip.h"

namespace mp2p_icp_hip {

// ================================================================== expressions
struct ExprCompiler;
namespace {
struct ExprParser {
  const std::string& s;
  const std::map<std::string, double>& vars;
  size_t i = 0;
  void ws() {
    while (i < s.size() && isspace((unsigned char)s[i])) i++;
  }
  [[noreturn]] void fail(const std::string& what) const {
    throw std::runtime_error("expression '" + s + "': " + what + " at position " + std::to_string(i));
  }
  double parse() {
    const double v = expr();
    ws();
    if (i != s.size()) fail("unexpected trailing characters");
    return v;
  }
  double expr() {
    double v = term();
    for (;;) {
      ws();
      if (i < s.size() && (s[i] == '+' || s[i] == '-')) {
        const char op = s[i++];
        const double r = term();
        v = op == '+' ? v + r : v - r;
      } else
        return v;
    }
  }
  double term() {
    double v = unary();
    for (;;) {
      ws();
      if (i < s.size() && (s[i] == '*' || s[i] == '/')) {
        const char op = s[i++];
        const double r = unary();
        v = op == '*' ? v * r : v / r;
      } else
        return v;
    }
  }
  double unary() {
    ws();
    if (i < s.size() && (s[i] == '-' || s[i] == '+')) {
      const char op = s[i++];
      const double r = unary();
      return op == '-' ? -r : r;
    }
    return power();
  }
  double power() {
    const double b = primary();
    ws();
    if (i < s.size() && s[i] == '^') {
      i++;
      return std::pow(b, unary());
    }
    return b;
  }
  double primary() {
    ws();
    if (i >= s.size()) fail("unexpected end");
    if (s[i] == '(') {
      i++;
      const double v = expr();
      ws();
      if (i >= s.size() || s[i] != ')') fail("missing ')'");
      i++;
      return v;
    }
    if (isdigit((unsigned char)s[i]) || s[i] == '.') {
      char* end = nullptr;
      const double v = strtod(s.c_str() + i, &end);
      i = end - s.c_str();
      return v;
    }
    if (isalpha((unsigned char)s[i]) || s[i] == '_') {
      size_t j = i;
      while (j < s.size() && (isalnum((unsigned char)s[j]) || s[j] == '_')) j++;
      const std::string id = s.substr(i, j - i);
      i = j;
      ws();
      if (i < s.size() && s[i] == '(') {
        i++;
        std::vector<double> a;
        ws();
        if (i < s.size() && s[i] == ')')
          i++;
        else
          for (;;) {
            a.push_back(expr());
            ws();
            if (i < s.size() && s[i] == ',') { i++; continue; }
            if (i < s.size() && s[i] == ')') { i++; break; }
            fail("missing ')' in call to " + id);
          }
        return call(id, a);
      }
      if (id == "pi" || id == "M_PI") return 3.14159265358979323846;
      if (id == "true") return 1.0;
      if (id == "false") return 0.0;
      auto it = vars.find(id);
      if (it == vars.end()) fail("unknown variable '" + id + "'");
      return it->second;
    }
    fail("unexpected character");
  }
  double call(const std::string& f, const std::vector<double>& a) const {
    auto need = [&](size_t n) {
      if (a.size() != n) throw std::runtime_error("expression '" + s + "': " + f + "() takes " + std::to_string(n) + " argument(s)");
    };
    if (f == "max") { if (a.empty()) need(1); double v = a[0]; for (double x : a) v = std::max(v, x); return v; }
    if (f == "min") { if (a.empty()) need(1); double v = a[0]; for (double x : a) v = std::min(v, x); return v; }
    if (f == "abs") { need(1); return std::fabs(a[0]); }
    if (f == "sqrt") { need(1); return std::sqrt(a[0]); }
    if (f == "exp") { need(1); return std::exp(a[0]); }
    if (f == "log") { need(1); return std::log(a[0]); }
    if (f == "sin") { need(1); return std::sin(a[0]); }
    if (f == "cos") { need(1); return std::cos(a[0]); }
    if (f == "tan") { need(1); return std::tan(a[0]); }
    if (f == "pow") { need(2); return std::pow(a[0], a[1]); }
    if (f == "clamp") { need(3); return std::min(std::max(a[0], a[1]), a[2]); }
    throw std::runtime_error("expression '" + s + "': unknown function '" + f + "'");
  }
};
}  // namespace

double evaluate_expression(const std::string& expr, const std::map<std::string, double>& vars) {
  ExprParser p{expr, vars};
  return p.parse();
}

// ---- the same grammar, emitting a stack program instead of a value
struct ExprCompiler {
  const std::string& s;
  CompiledExpression& out;
  size_t i = 0;
  enum { CONST = 0, VAR, ADD, SUB, MUL, DIV, NEG, POW, CALL };
  enum Fn { F_MAX = 0, F_MIN, F_ABS, F_SQRT, F_EXP, F_LOG, F_SIN, F_COS, F_TAN, F_POW, F_CLAMP };
  void ws() { while (i < s.size() && isspace((unsigned char)s[i])) i++; }
  [[noreturn]] void fail(const std::string& what) const {
    throw std::runtime_error("expression '" + s + "': " + what + " at position " + std::to_string(i));
  }
  void emit(int code, int arg = 0, int nargs = 0, double v = 0) { out.prog_.push_back({code, arg, nargs, v}); }
  void compile() {
    expr();
    ws();
    if (i != s.size()) fail("unexpected trailing characters");
  }
  void expr() {
    term();
    for (;;) {
      ws();
      if (i < s.size() && (s[i] == '+' || s[i] == '-')) {
        const char op = s[i++];
        term();
        emit(op == '+' ? ADD : SUB);
      } else
        return;
    }
  }
  void term() {
    unary();
    for (;;) {
      ws();
      if (i < s.size() && (s[i] == '*' || s[i] == '/')) {
        const char op = s[i++];
        unary();
        emit(op == '*' ? MUL : DIV);
      } else
        return;
    }
  }
  void unary() {
    ws();
    if (i < s.size() && (s[i] == '-' || s[i] == '+')) {
      const char op = s[i++];
      unary();
      if (op == '-') emit(NEG);
      return;
    }
    power();
  }
  void power() {
    primary();
    ws();
    if (i < s.size() && s[i] == '^') {
      i++;
      unary();
      emit(POW);
    }
  }
  void primary() {
    ws();
    if (i >= s.size()) fail("unexpected end");
    if (s[i] == '(') {
      i++;
      expr();
      ws();
      if (i >= s.size() || s[i] != ')') fail("missing ')'");
      i++;
      return;
    }
    if (isdigit((unsigned char)s[i]) || s[i] == '.') {
      char* end = nullptr;
      const double v = strtod(s.c_str() + i, &end);
      i = end - s.c_str();
      emit(CONST, 0, 0, v);
      return;
    }
    if (isalpha((unsigned char)s[i]) || s[i] == '_') {
      size_t j = i;
      while (j < s.size() && (isalnum((unsigned char)s[j]) || s[j] == '_')) j++;
      const std::string id = s.substr(i, j - i);
      i = j;
      ws();
      if (i < s.size() && s[i] == '(') {
        i++;
        int n = 0;
        ws();
        if (i < s.size() && s[i] == ')')
          i++;
        else
          for (;;) {
            expr();
            n++;
            ws();
            if (i < s.size() && s[i] == ',') { i++; continue; }
            if (i < s.size() && s[i] == ')') { i++; break; }
            fail("missing ')' in call to " + id);
          }
        static const char* names[] = {"max", "min", "abs", "sqrt", "exp", "log", "sin", "cos", "tan", "pow", "clamp"};
        static const int arity[] = {-1, -1, 1, 1, 1, 1, 1, 1, 1, 2, 3};
        for (int f = 0; f < 11; f++)
          if (id == names[f]) {
            if ((arity[f] >= 0 && n != arity[f]) || (arity[f] < 0 && n < 1))
              throw std::runtime_error("expression '" + s + "': " + id + "() called with " + std::to_string(n) + " argument(s)");
            emit(CALL, f, n);
            return;
          }
        throw std::runtime_error("expression '" + s + "': unknown function '" + id + "'");
      }
      if (id == "pi" || id == "M_PI") { emit(CONST, 0, 0, 3.14159265358979323846); return; }
      if (id == "true") { emit(CONST, 0, 0, 1.0); return; }
      if (id == "false") { emit(CONST, 0, 0, 0.0); return; }
      int idx = -1;
      for (size_t k = 0; k < out.vars_.size(); k++)
        if (out.vars_[k] == id) idx = (int)k;
      if (idx < 0) {
        idx = (int)out.vars_.size();
        out.vars_.push_back(id);
      }
      emit(VAR, idx);
      return;
    }
    fail("unexpected character");
  }
};

CompiledExpression::CompiledExpression(const std::string& text) : text_(text) {
  ExprCompiler c{text_, *this};
  c.compile();
}

double CompiledExpression::evaluate(const std::vector<const double*>& values) const {
  double st[32];
  int sp = 0;
  for (const Op& o : prog_) {
    switch (o.code) {
      case ExprCompiler::CONST: st[sp++] = o.value; break;
      case ExprCompiler::VAR: st[sp++] = *values[o.arg]; break;
      case ExprCompiler::ADD: sp--; st[sp - 1] = st[sp - 1] + st[sp]; break;
      case ExprCompiler::SUB: sp--; st[sp - 1] = st[sp - 1] - st[sp]; break;
      case ExprCompiler::MUL: sp--; st[sp - 1] = st[sp - 1] * st[sp]; break;
      case ExprCompiler::DIV: sp--; st[sp - 1] = st[sp - 1] / st[sp]; break;
      case ExprCompiler::NEG: st[sp - 1] = -st[sp - 1]; break;
      case ExprCompiler::POW: sp--; st[sp - 1] = std::pow(st[sp - 1], st[sp]); break;
      default: {
        double* a = &st[sp - o.nargs];
        double v = a[0];
        switch (o.arg) {
          case ExprCompiler::F_MAX: for (int k = 1; k < o.nargs; k++) v = std::max(v, a[k]); break;
          case ExprCompiler::F_MIN: for (int k = 1; k < o.nargs; k++) v = std::min(v, a[k]); break;
          case ExprCompiler::F_ABS: v = std::fabs(a[0]); break;
          case ExprCompiler::F_SQRT: v = std::sqrt(a[0]); break;
          case ExprCompiler::F_EXP: v = std::exp(a[0]); break;
          case ExprCompiler::F_LOG: v = std::log(a[0]); break;
          case ExprCompiler::F_SIN: v = std::sin(a[0]); break;
          case ExprCompiler::F_COS: v = std::cos(a[0]); break;
          case ExprCompiler::F_TAN: v = std::tan(a[0]); break;
          case ExprCompiler::F_POW: v = std::pow(a[0], a[1]); break;
          case ExprCompiler::F_CLAMP: v = std::min(std::max(a[0], a[1]), a[2]); break;
        }
        sp -= o.nargs;
        st[sp++] = v;
      }
    }
    if (sp >= 31) throw std::runtime_error("expression '" + text_ + "': too deeply nested");
  }
  return st[0];
}

double CompiledExpression::evaluate(const std::map<std::string, double>& vars) const {
  std::vector<const double*> vals(vars_.size());
  for (size_t k = 0; k < vars_.size(); k++) {
    auto it = vars.find(vars_[k]);
    if (it == vars.end()) throw std::runtime_error("expression '" + text_ + "': unknown variable '" + vars_[k] + "'");
    vals[k] = &it->second;
  }
  return evaluate(vals);
}

void ParameterSource::realize() {
  for (auto* p : attached_) p->realizeWith(vars_);
}

void Parameterizable::realizeWith(const std::map<std::string, double>& vars) {
  for (auto& d : declared_) *d.target = d.compiled->evaluate(vars);
}

Parameterizable::Binding Parameterizable::bind(const std::map<std::string, double>& vars) const {
  Binding b;
  for (const auto& d : declared_) {
    std::vector<const double*> vals;
    for (const auto& name : d.compiled->variables()) {
      auto it = vars.find(name);
      if (it == vars.end()) throw std::runtime_error("expression '" + d.expr + "': unknown variable '" + name + "'");
      vals.push_back(&it->second);
    }
    b.items.emplace_back(&d, std::move(vals));
  }
  return b;
}

void Parameterizable::Binding::realize() const {
  for (const auto& it : items) *it.first->target = it.first->compiled->evaluate(it.second);
}

static bool looks_numeric(const std::string& s) {
  if (s.empty()) return false;
  char* end = nullptr;
  strtod(s.c_str(), &end);
  while (end && *end && isspace((unsigned char)*end)) end++;
  return end && *end == '\0';
}

void Parameterizable::parameterFromConfig(const Config& c, const std::string& name, double* target, bool required) {
  if (!c.has(name)) {
    if (required) throw std::runtime_error("missing required parameter '" + name + "'");
    return;
  }
  const std::string v = c[name].asString();
  if (looks_numeric(v))
    *target = strtod(v.c_str(), nullptr);
  else
    declareParameter(name, v, target);  // a formula: evaluated at realize()
}

// ================================================================== YAML subset
namespace {
std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && isspace((unsigned char)s[a])) a++;
  while (b > a && isspace((unsigned char)s[b - 1])) b--;
  return s.substr(a, b - a);
}

std::string strip_comment(const std::string& line) {
  bool sq = false, dq = false;
  for (size_t i = 0; i < line.size(); i++) {
    const char c = line[i];
    if (c == '\'' && !dq) sq = !sq;
    if (c == '"' && !sq) dq = !dq;
    if (c == '#' && !sq && !dq && (i == 0 || isspace((unsigned char)line[i - 1]))) return line.substr(0, i);
  }
  return line;
}

std::string unquote(const std::string& s) {
  if (s.size() >= 2 && ((s.front() == '\'' && s.back() == '\'') || (s.front() == '"' && s.back() == '"')))
    return s.substr(1, s.size() - 2);
  return s;
}

// ${VAR|default} (mola_yaml [U]); $f{...} arithmetic is evaluated with the formula evaluator
std::string substitute_env(std::string s) {
  for (int guard = 0; guard < 64; guard++) {
    const size_t a = s.find("${");
    if (a == std::string::npos) break;
    int depth = 0;
    size_t b = a + 1;
    for (; b < s.size(); b++) {
      if (s[b] == '{') depth++;
      if (s[b] == '}' && --depth == 0) break;
    }
    if (b >= s.size()) break;
    const std::string body = s.substr(a + 2, b - a - 2);
    const size_t bar = body.find('|');
    const std::string var = bar == std::string::npos ? body : body.substr(0, bar);
    const char* env = getenv(var.c_str());
    std::string val;
    if (env)
      val = env;
    else if (bar != std::string::npos)
      val = body.substr(bar + 1);
    else
      throw std::runtime_error("environment variable '" + var + "' is not set and has no default");
    s = s.substr(0, a) + val + s.substr(b + 1);
  }
  size_t from = 0;
  for (int guard = 0; guard < 64; guard++) {
    const size_t a = s.find("$f{", from);
    if (a == std::string::npos) break;
    const size_t b = s.find('}', a);
    if (b == std::string::npos) break;
    try {
      std::ostringstream os;
      os.precision(17);
      os << evaluate_expression(s.substr(a + 3, b - a - 3), {});
      s = s.substr(0, a) + os.str() + s.substr(b + 1);
    } catch (const std::exception&) {
      from = b + 1;  // refers to run-time variables (e.g. ESTIMATED_SENSOR_MAX_RANGE): left for the consumer
    }
  }
  return s;
}

struct Line {
  int indent;
  std::string text;
};

Config parse_flow_map(const std::string& t) {  // {a: b, c: "d"}
  Config c;
  c.kind = Config::Kind::Map;
  const std::string body = trim(t.substr(1, t.size() - 2));
  size_t i = 0;
  while (i < body.size()) {
    size_t j = i;
    bool sq = false, dq = false;
    while (j < body.size() && (sq || dq || body[j] != ',')) {
      if (body[j] == '\'' && !dq) sq = !sq;
      if (body[j] == '"' && !sq) dq = !dq;
      j++;
    }
    const std::string item = trim(body.substr(i, j - i));
    const size_t colon = item.find(':');
    if (colon == std::string::npos) throw std::runtime_error("YAML: bad flow-map item '" + item + "'");
    Config v;
    v.kind = Config::Kind::Scalar;
    v.scalar = substitute_env(unquote(trim(item.substr(colon + 1))));
    c.map.emplace_back(unquote(trim(item.substr(0, colon))), v);
    i = j + 1;
  }
  return c;
}

Config parse_scalar_or_flow(const std::string& t);

Config parse_flow_seq(const std::string& t) {  // [a, 'b', max(1, c)]: commas inside quotes / brackets do not split
  Config c;
  c.kind = Config::Kind::Seq;
  const std::string body = trim(t.substr(1, t.size() - 2));
  size_t i = 0;
  while (i < body.size()) {
    size_t j = i;
    bool sq = false, dq = false;
    int depth = 0;
    while (j < body.size() && (sq || dq || depth > 0 || body[j] != ',')) {
      if (body[j] == '\'' && !dq) sq = !sq;
      if (body[j] == '"' && !sq) dq = !dq;
      if (!sq && !dq && (body[j] == '(' || body[j] == '{' || body[j] == '[')) depth++;
      if (!sq && !dq && (body[j] == ')' || body[j] == '}' || body[j] == ']')) depth--;
      j++;
    }
    const std::string item = trim(body.substr(i, j - i));
    if (!item.empty()) c.seq.push_back(parse_scalar_or_flow(item));
    i = j + 1;
  }
  return c;
}

Config parse_scalar_or_flow(const std::string& t) {
  if (t.size() >= 2 && t.front() == '{' && t.back() == '}') return parse_flow_map(t);
  if (t.size() >= 2 && t.front() == '[' && t.back() == ']') return parse_flow_seq(t);
  Config c;
  c.kind = Config::Kind::Scalar;
  c.scalar = substitute_env(unquote(t));
  return c;
}

size_t find_key_colon(const std::string& t) {  // first ':' outside quotes followed by space/end
  bool sq = false, dq = false;
  for (size_t i = 0; i < t.size(); i++) {
    if (t[i] == '\'' && !dq) sq = !sq;
    if (t[i] == '"' && !sq) dq = !dq;
    if (t[i] == ':' && !sq && !dq && (i + 1 == t.size() || isspace((unsigned char)t[i + 1]))) return i;
  }
  return std::string::npos;
}

Config parse_block(const std::vector<Line>& L, size_t& i, int indent);

Config parse_value_after_key(const std::vector<Line>& L, size_t& i, int key_indent, const std::string& rest) {
  if (!rest.empty()) return parse_scalar_or_flow(rest);
  if (i < L.size() && (L[i].indent > key_indent || (L[i].indent == key_indent && L[i].text.rfind("- ", 0) == 0))) {
    const std::string& t = L[i].text;
    if (t.rfind("- ", 0) != 0 && t != "-" && t.front() != '{' && find_key_colon(t) == std::string::npos)
      return parse_scalar_or_flow(L[i++].text);  // a plain scalar on its own line (e.g. "~")
    return parse_block(L, i, L[i].indent);
  }
  return Config{};  // null
}

Config parse_block(const std::vector<Line>& L, size_t& i, int indent) {
  Config c;
  if (i >= L.size()) return c;
  if (L[i].text.rfind("- ", 0) == 0 || L[i].text == "-") {
    c.kind = Config::Kind::Seq;
    while (i < L.size() && L[i].indent == indent && (L[i].text.rfind("- ", 0) == 0 || L[i].text == "-")) {
      const std::string item = trim(L[i].text.substr(1));
      const int item_indent = indent + 2;
      if (item.empty()) {
        i++;
        c.seq.push_back(i < L.size() && L[i].indent > indent ? parse_block(L, i, L[i].indent) : Config{});
      } else if (item.front() == '{') {
        c.seq.push_back(parse_flow_map(item));
        i++;
      } else if (find_key_colon(item) != std::string::npos) {
        // "- key: value" starts a map whose further keys are indented to the key's column
        std::vector<Line> sub;
        sub.push_back({item_indent, item});
        size_t j = i + 1;
        while (j < L.size() && L[j].indent >= item_indent) sub.push_back(L[j++]);
        size_t k = 0;
        c.seq.push_back(parse_block(sub, k, item_indent));
        i = j;
      } else {
        c.seq.push_back(parse_scalar_or_flow(item));
        i++;
      }
    }
    return c;
  }
  c.kind = Config::Kind::Map;
  while (i < L.size() && L[i].indent == indent) {
    const std::string& t = L[i].text;
    const size_t colon = find_key_colon(t);
    if (colon == std::string::npos) throw std::runtime_error("YAML: expected 'key: value' but got '" + t + "'");
    const std::string key = unquote(trim(t.substr(0, colon)));
    const std::string rest = trim(t.substr(colon + 1));
    i++;
    c.map.emplace_back(key, parse_value_after_key(L, i, indent, rest));
  }
  return c;
}
}  // namespace

Config Config::FromYamlText(const std::string& text) {
  std::vector<Line> lines;
  std::istringstream is(text);
  std::string raw;
  while (std::getline(is, raw)) {
    const std::string nc = strip_comment(raw);
    if (trim(nc).empty()) continue;
    int ind = 0;
    while (ind < (int)nc.size() && nc[ind] == ' ') ind++;
    lines.push_back({ind, trim(nc)});
  }
  size_t i = 0;
  if (lines.empty()) return Config{};
  return parse_block(lines, i, lines[0].indent);
}

Config Config::FromYamlFile(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open '" + path + "'");
  std::stringstream ss;
  ss << f.rdbuf();
  return FromYamlText(ss.str());
}

bool Config::has(const std::string& key) const {
  for (auto& kv : map)
    if (kv.first == key) return true;
  return false;
}

const Config& Config::operator[](const std::string& key) const {
  for (auto& kv : map)
    if (kv.first == key) return kv.second;
  throw std::runtime_error("missing key '" + key + "'");
}

std::string Config::getOr(const std::string& key, const std::string& def) const {
  return has(key) && !(*this)[key].isNull() ? (*this)[key].asString() : def;
}

}  // namespace mp2p_icp_hip
