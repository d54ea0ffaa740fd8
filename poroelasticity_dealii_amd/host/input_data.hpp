// Parameter file front end: the same sections / keys / defaults / unit conversions as
// input_data::InputDataPoroel (lib/include/InputDataPoroel.h:89-222), parsed from the
// deal.II ParameterHandler text grammar (`subsection X` / `set key = value` / `end`, `#` comments),
// so the bundled input.data drives this build unchanged.
#pragma once
#include <cctype>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/poroel_hip.h"

namespace input_data {

inline std::string trim(const std::string &s) {
  size_t a = 0, b = s.size();
  while (a < b && std::isspace((unsigned char)s[a])) ++a;
  while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
  return s.substr(a, b - a);
}

// InputDataPoroel.h:9-25
template <typename T>
std::vector<T> parse_string_list(std::string list_string, char delimiter = ',') {
  std::vector<T> list;
  if (trim(list_string).empty()) return list;
  std::stringstream ss(list_string);
  std::string item;
  while (std::getline(ss, item, delimiter)) {
    std::stringstream convert(item);
    T v; convert >> v; list.push_back(v);
  }
  return list;
}

class InputDataPoroel {
 public:
  // mesh
  int dim = 2;
  std::vector<double> domain_size{10, 10};
  int initial_refinement_level = 3, max_refinement_level = 5;
  // equation data
  double perm = 1 * 9.869233e-16, poro = 0.3, visc = 1e-3, f_comp = 45.8e-11;
  double youngs_modulus = 7e9, poisson_ratio = 0.3, biot_coef = 0.9;
  double bulk_density = 2700, r_well = 0.1, flow_rate = 1e-6;
  // solver control
  double time_step = 60, t_max = 60, fss_tol = 1e-8, pressure_tol = 1e-8;
  int max_fss_iterations = 50, max_pressure_iterations = 50;
  // in situ
  double p_init = 10e6;
  std::vector<int> stress_boundary_labels, stress_boundary_components;
  std::vector<double> stress_boundary_values;
  std::vector<int> displacement_boundary_labels{0, 2, 3, 1}, displacement_boundary_components{1, 1, 0, 0};
  std::vector<double> displacement_boundary_values{0, 0, 0, -0.1};
  // derived (:213-222)
  double lame_constant = 0, shear_modulus = 0, bulk_modulus = 0, grain_bulk_modulus = 0, n_modulus = 0, m_modulus = 0;

  InputDataPoroel() { compute_derived_parameters(); }

  void read_input_file(const std::string &input_file_name) {
    std::ifstream in(input_file_name);
    if (!in) throw std::runtime_error("cannot open parameter file " + input_file_name);
    std::map<std::string, std::string> kv;  // "Section/key" -> value
    std::string line, section;
    while (std::getline(in, line)) {
      const size_t hash = line.find('#');
      if (hash != std::string::npos) line = line.substr(0, hash);
      line = trim(line);
      if (line.empty()) continue;
      if (line.rfind("subsection", 0) == 0) section = trim(line.substr(10));
      else if (line == "end") section.clear();
      else if (line.rfind("set", 0) == 0) {
        const size_t eq = line.find('=');
        if (eq == std::string::npos) throw std::runtime_error("parameter file: missing '=' in: " + line);
        kv[section + "/" + trim(line.substr(3, eq - 3))] = trim(line.substr(eq + 1));
      } else throw std::runtime_error("parameter file: cannot parse: " + line);
    }
    assign_parameters(kv);
    compute_derived_parameters();
  }

  void compute_derived_parameters() {  // :213-222
    const double E = youngs_modulus, nu = poisson_ratio;
    lame_constant = E * nu / ((1. + nu) * (1. - 2. * nu));
    shear_modulus = 0.5 * E / (1 + nu);
    bulk_modulus = lame_constant + 2. / 3. * shear_modulus;
    grain_bulk_modulus = bulk_modulus / (1. - biot_coef);
    n_modulus = grain_bulk_modulus / (biot_coef - poro);
    m_modulus = (n_modulus / f_comp) / (n_modulus * poro + 1. / f_comp);
  }

  poro_material material() const {
    poro_material m{};
    m.lame_lambda = lame_constant; m.shear_G = shear_modulus; m.biot_alpha = biot_coef;
    m.bulk_K = bulk_modulus; m.biot_M = m_modulus; m.k_over_mu = perm / visc;
    m.r_well = r_well; m.flow_rate = flow_rate;
    return m;
  }

 private:
  static void range(const std::string &key, double v, double lo, double hi) {
    if (v < lo || v > hi) throw std::runtime_error("parameter '" + key + "' out of range");
  }
  void assign_parameters(const std::map<std::string, std::string> &kv) {  // :150-210, ranges :93-141
    auto has = [&](const char *k) { return kv.find(k) != kv.end(); };
    auto num = [&](const char *k, double &dst, double lo, double hi) {
      if (!has(k)) return;
      dst = std::stod(kv.at(k)); range(k, dst, lo, hi);
    };
    auto integer = [&](const char *k, int &dst, int lo, int hi) {
      if (!has(k)) return;
      dst = std::stoi(kv.at(k)); range(k, dst, lo, hi);
    };
    const double inf = 1e300;
    integer("Mesh/Dimensions", dim, 1, 3);
    if (has("Mesh/Domain size")) domain_size = parse_string_list<double>(kv.at("Mesh/Domain size"));
    integer("Mesh/Initial refinement level", initial_refinement_level, 2, 1 << 30);
    integer("Mesh/Max refinement level", max_refinement_level, 2, 1 << 30);
    const double mili_darcy = 9.869233e-16;  // :162
    num("Properties/Young modulus", youngs_modulus, 1, inf);
    num("Properties/Poisson ratio", poisson_ratio, 0, 0.5);
    num("Properties/Biot coefficient", biot_coef, 0.1, 1);
    if (has("Properties/Permeability")) { double k = 0; num("Properties/Permeability", k, 1e-20, 1e5); perm = k * mili_darcy; }
    num("Properties/Porosity", poro, 1e-5, 0.99999);
    num("Properties/Viscosity", visc, 1e-6, 1);
    num("Properties/Bulk density", bulk_density, 5e2, 1e4);
    num("Properties/Fluid compressibility", f_comp, 1e-16, 1e-2);
    num("Properties/Well radius", r_well, 1e-2, inf);
    num("Properties/Flow rate", flow_rate, -inf, inf);
    num("In situ/Initial pressure", p_init, 0, inf);
    if (has("In situ/Stress boundary labels")) stress_boundary_labels = parse_string_list<int>(kv.at("In situ/Stress boundary labels"));
    if (has("In situ/Stress boundary components")) stress_boundary_components = parse_string_list<int>(kv.at("In situ/Stress boundary components"));
    if (has("In situ/Stress boundary values")) stress_boundary_values = parse_string_list<double>(kv.at("In situ/Stress boundary values"));
    if (has("In situ/Displacement boundary labels")) displacement_boundary_labels = parse_string_list<int>(kv.at("In situ/Displacement boundary labels"));
    if (has("In situ/Displacement boundary components")) displacement_boundary_components = parse_string_list<int>(kv.at("In situ/Displacement boundary components"));
    if (has("In situ/Displacement boundary values")) displacement_boundary_values = parse_string_list<double>(kv.at("In situ/Displacement boundary values"));
    num("Solver/Time step", time_step, 1e-8, inf);
    num("Solver/Time max", t_max, 1e-8, inf);
    integer("Solver/Max FSS iterations", max_fss_iterations, 1, 1000);
    integer("Solver/Max pressure iterations", max_pressure_iterations, 1, 1000);
    num("Solver/FSS tolerance", fss_tol, 1e-20, 1e-1);
    num("Solver/Pressure tolerance", pressure_tol, 1e-20, 1e-1);
  }
};

}  // namespace input_data
