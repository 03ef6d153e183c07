// wavehip_mesh.hpp -- C++ host side of the mesh-FILE path: what the reference's driver does with
// DOLFINx before it builds LinearGLLOpt (demo/cpu_planar3d/main.cpp:36-66):
//   io::XDMFFile("mesh.xdmf").read_mesh(element, ghost_mode, "planar3d")      -> read_mesh
//   read_meshtags(mesh, "planar3d_boundaries")                                -> read_mesh(.., tags)
//   fem::create_functionspace(mesh, Lagrange(hexahedron, degree, gll_warped)) -> create_functionspace
//   the tagged boundary dof sets of the form L (forms.ufl:19-24, LinearGLL.hpp:113-115) -> boundary_set
//   mesh::h and the CFL rule (main.cpp:48-66)                                 -> cfl_time_step
// Thin wrappers over the host-only C ABI (wf_mesh_*, wf_fs_*; csrc/mesh_io.cpp, csrc/function_space.cpp),
// shared with the Python package (mesh_io.py).  Hexahedral meshes with cells in any local orientation.
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "wavehip.hpp"

namespace wavehip {

struct FileMesh {
  std::vector<double> x;              // [nverts][3]
  std::vector<std::int32_t> cells;    // [ncells][8], vertex v = a + 2b + 4c
  std::int64_t nverts() const { return (std::int64_t)x.size() / 3; }
  std::int64_t ncells() const { return (std::int64_t)cells.size() / 8; }
};

/// mesh::MeshTags<int32> of dimension 2: the four vertices (tensor order) of every tagged facet and its value
struct MeshTags {
  std::vector<std::int32_t> facet_vertices;   // [nfacets][4]
  std::vector<std::int32_t> values;           // [nfacets]
};

namespace detail {
struct MeshFileHandle {
  wf_mesh_file* h = nullptr;
  MeshFileHandle(const std::string& xdmf, const std::string& grid) { check(wf_mesh_open(xdmf.c_str(), grid.c_str(), &h)); }
  ~MeshFileHandle() { wf_mesh_close(h); }
};
inline void read_into(MeshFileHandle& f, FileMesh& m)
{
  std::int64_t nv = 0, nc = 0;
  check(wf_mesh_sizes(f.h, &nv, &nc));
  m.x.resize((std::size_t)nv * 3);
  m.cells.resize((std::size_t)nc * 8);
  check(wf_mesh_read(f.h, m.x.data(), m.cells.data()));
}
}  // namespace detail

/// XDMFFile(path).read_mesh(..., grid_name)
inline FileMesh read_mesh(const std::string& xdmf_path, const std::string& grid_name)
{
  detail::MeshFileHandle f(xdmf_path, grid_name);
  FileMesh m;
  detail::read_into(f, m);
  return m;
}

/// read_mesh + read_meshtags(mesh, tags_name)
inline std::pair<FileMesh, MeshTags> read_mesh(const std::string& xdmf_path, const std::string& grid_name,
                                               const std::string& tags_name)
{
  detail::MeshFileHandle f(xdmf_path, grid_name);
  std::pair<FileMesh, MeshTags> out;
  detail::read_into(f, out.first);
  std::int64_t nf = 0;
  check(wf_mesh_tags_size(f.h, tags_name.c_str(), &nf));
  out.second.facet_vertices.resize((std::size_t)nf * 4);
  out.second.values.resize((std::size_t)nf);
  check(wf_mesh_read_tags(f.h, tags_name.c_str(), out.second.facet_vertices.data(), out.second.values.data()));
  return out;
}

/// The data of fem::create_functionspace for a FileMesh: dofs identified topologically, numbered in
/// lexicographic (z, y, x) order of their coordinates; tensor-ordered dofmap in each cell's own frame.
struct MeshSpace {
  const FileMesh* mesh = nullptr;
  int degree = 0;
  std::int64_t ndofs = 0;
  std::vector<std::int32_t> dofmap;   // [ncells][(degree+1)^3]
  Space space() const
  {
    Space S;
    S.degree = degree;
    S.ncells = (std::int32_t)mesh->ncells();
    S.ndofs = (std::int32_t)ndofs;
    S.dofmap = dofmap.data();
    S.nverts = (std::int32_t)mesh->nverts();
    S.x = mesh->x.data();
    S.geom_dofmap = mesh->cells.data();
    return S;
  }
};

// Setup-time renumbering option (wf_lattice_numbering): renumber the space so that its dofs follow the lattice
// columns the marching kernels walk; returns new_of_old (vectors of the old numbering move as
// x_new[new_of_old[d]] = x_old[d]).  DOLFINx's own numbering is local enough (first-touch: 0.222 vs 0.208 ms at
// cfg2); a scattered one costs 2.8x and is brought back to 0.217 ms by this.
inline std::vector<std::int32_t> renumber_lattice(MeshSpace& V)
{
  std::vector<std::int32_t> new_of_old((std::size_t)V.ndofs);
  check(wf_lattice_numbering(V.degree, (std::int64_t)V.mesh->ncells(), (std::int32_t)V.ndofs, V.dofmap.data(), new_of_old.data()));
  for (auto& d : V.dofmap) d = new_of_old[(std::size_t)d];
  return new_of_old;
}

inline MeshSpace create_functionspace(const FileMesh& mesh, int degree)
{
  MeshSpace V;
  V.mesh = &mesh;
  V.degree = degree;
  const std::size_t nd = (std::size_t)(degree + 1) * (degree + 1) * (degree + 1);
  V.dofmap.resize((std::size_t)mesh.ncells() * nd);
  check(wf_fs_build(degree, mesh.nverts(), mesh.x.data(), mesh.ncells(), mesh.cells.data(), &V.ndofs, V.dofmap.data(), nullptr, 0));
  return V;
}

/// The boundary dof set of the facets tagged `value` with its collocated facet masses (diagonal GLL form of
/// inner(g, v) * ds(value), forms.ufl:19-24): (ascending dof indices, masses)
inline std::pair<std::vector<std::int32_t>, std::vector<double>> boundary_set(const MeshSpace& V, const MeshTags& tags, int value)
{
  std::vector<std::int32_t> fv;
  for (std::size_t f = 0; f < tags.values.size(); ++f)
    if (tags.values[f] == value) fv.insert(fv.end(), tags.facet_vertices.begin() + 4 * f, tags.facet_vertices.begin() + 4 * f + 4);
  const std::int64_t nf = (std::int64_t)fv.size() / 4;
  std::vector<std::int32_t> cell((std::size_t)nf), axis((std::size_t)nf), side((std::size_t)nf);
  check(wf_fs_locate_facets(V.mesh->ncells(), V.mesh->cells.data(), nf, fv.data(), cell.data(), axis.data(), side.data()));
  const std::size_t cap = (std::size_t)nf * (V.degree + 1) * (V.degree + 1) + 1;
  std::pair<std::vector<std::int32_t>, std::vector<double>> out;
  out.first.resize(cap);
  out.second.resize(cap);
  std::int64_t n = 0;
  check(wf_fs_facet_mass(V.degree, V.mesh->nverts(), V.mesh->x.data(), V.mesh->ncells(), V.mesh->cells.data(), V.dofmap.data(), nf,
                         cell.data(), axis.data(), side.data(), &n, out.first.data(), out.second.data()));
  out.first.resize((std::size_t)n);
  out.second.resize((std::size_t)n);
  return out;
}

/// demo/cpu_planar3d/main.cpp:48-66: (time step, steps per period) from mesh::h = the smallest cell diameter
inline std::pair<double, int> cfl_time_step(const FileMesh& mesh, int degree, double c0, double freq, double CFL = 0.5)
{
  double h = 0.0;
  check(wf_fs_min_cell_diameter(mesh.nverts(), mesh.x.data(), mesh.ncells(), mesh.cells.data(), &h));
  const double dt = CFL * h / (c0 * degree * degree), period = 1.0 / freq;
  const int spp = (int)(period / dt + 1);
  return {period / spp, spp};
}

}  // namespace wavehip
