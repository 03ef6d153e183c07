// Mesh / facet-tag input in the format the reference reads:
// demo/cpu_planar3d/main.cpp:39-45 -- io::XDMFFile("../mesh.xdmf").read_mesh(..., "planar3d")
// and read_meshtags(mesh, "planar3d_boundaries"): an XDMF (XML) file whose DataItems point
// into an HDF5 file,
//   /Mesh/<name>/topology        [ncells][8]   hexahedra, XDMF vertex order
//   /Mesh/<name>/geometry        [nverts][3]
//   /MeshTags/<tags>/topology    [nfacets][4]  quadrilaterals, XDMF vertex order
//   /MeshTags/<tags>/Values      [nfacets]
// (the paths are taken from the XML, so files written by other tools work as long as the
// grids are named).  The reference's mesh.xdmf is not in its repository; parity of this
// reader is pinned by round trips through the writer below, not by a reference file.
//
// Host-only code.  libhdf5 is bound at run time (dlopen), like librccl: libwavehip has no
// hard dependency on it.  Vertex orders are converted to the engine's tensor order
// (vertex v = a + 2b + 4c, wavehip.h): XDMF hexahedron (0,1,2,3,4,5,6,7) -> (0,1,3,2,4,5,7,6),
// quadrilateral (0,1,2,3) -> (0,1,3,2).
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>

#include "common.h"

namespace {

typedef int64_t hid_t;
typedef int herr_t;
typedef unsigned long long hsize_t;

struct H5 {
  void* handle = nullptr;
  herr_t (*open)() = nullptr;
  hid_t (*Fopen)(const char*, unsigned, hid_t) = nullptr;
  hid_t (*Fcreate)(const char*, unsigned, hid_t, hid_t) = nullptr;
  herr_t (*Fclose)(hid_t) = nullptr;
  hid_t (*Dopen2)(hid_t, const char*, hid_t) = nullptr;
  hid_t (*Dcreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t, hid_t) = nullptr;
  hid_t (*Dget_space)(hid_t) = nullptr;
  herr_t (*Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void*) = nullptr;
  herr_t (*Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void*) = nullptr;
  herr_t (*Dclose)(hid_t) = nullptr;
  int (*Sget_simple_extent_ndims)(hid_t) = nullptr;
  int (*Sget_simple_extent_dims)(hid_t, hsize_t*, hsize_t*) = nullptr;
  hid_t (*Screate_simple)(int, const hsize_t*, const hsize_t*) = nullptr;
  herr_t (*Sclose)(hid_t) = nullptr;
  hid_t (*Gcreate2)(hid_t, const char*, hid_t, hid_t, hid_t) = nullptr;
  herr_t (*Gclose)(hid_t) = nullptr;
  hid_t (*Pcreate)(hid_t) = nullptr;
  herr_t (*Pset_create_intermediate_group)(hid_t, unsigned) = nullptr;
  herr_t (*Pclose)(hid_t) = nullptr;
  herr_t (*Eset_auto2)(hid_t, void*, void*) = nullptr;
  hid_t NATIVE_DOUBLE = -1, NATIVE_INT64 = -1, NATIVE_INT32 = -1, P_LINK_CREATE = -1;
};
H5* g_h5 = nullptr;

template <typename F>
bool bind(void* h, const char* name, F* fn)
{
  *fn = reinterpret_cast<F>(dlsym(h, name));
  return *fn != nullptr;
}

H5* h5()
{
  if (g_h5) return g_h5;
  std::vector<std::string> cands;
  if (const char* e = std::getenv("WF_HDF5_LIB")) cands.push_back(e);
  cands.push_back("libhdf5.so");
  cands.push_back("libhdf5_serial.so");
  cands.push_back("/opt/conda/lib/libhdf5.so");
  std::string tried;
  for (const auto& c : cands) {
    void* h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      tried += c + "; ";
      continue;
    }
    auto a = std::make_unique<H5>();
    a->handle = h;
    bool ok = bind(h, "H5open", &a->open) && bind(h, "H5Fopen", &a->Fopen) && bind(h, "H5Fcreate", &a->Fcreate)
              && bind(h, "H5Fclose", &a->Fclose) && bind(h, "H5Dopen2", &a->Dopen2) && bind(h, "H5Dcreate2", &a->Dcreate2)
              && bind(h, "H5Dget_space", &a->Dget_space) && bind(h, "H5Dread", &a->Dread) && bind(h, "H5Dwrite", &a->Dwrite)
              && bind(h, "H5Dclose", &a->Dclose) && bind(h, "H5Sget_simple_extent_ndims", &a->Sget_simple_extent_ndims)
              && bind(h, "H5Sget_simple_extent_dims", &a->Sget_simple_extent_dims)
              && bind(h, "H5Screate_simple", &a->Screate_simple) && bind(h, "H5Sclose", &a->Sclose)
              && bind(h, "H5Pcreate", &a->Pcreate) && bind(h, "H5Pset_create_intermediate_group", &a->Pset_create_intermediate_group)
              && bind(h, "H5Pclose", &a->Pclose) && bind(h, "H5Eset_auto2", &a->Eset_auto2);
    if (ok && a->open() >= 0) {
      auto gv = [&](const char* n) {
        hid_t* p = reinterpret_cast<hid_t*>(dlsym(h, n));
        return p ? *p : (hid_t)-1;
      };
      a->NATIVE_DOUBLE = gv("H5T_NATIVE_DOUBLE_g");
      a->NATIVE_INT64 = gv("H5T_NATIVE_INT64_g");
      a->NATIVE_INT32 = gv("H5T_NATIVE_INT32_g");
      a->P_LINK_CREATE = gv("H5P_CLS_LINK_CREATE_ID_g");
      ok = a->NATIVE_DOUBLE >= 0 && a->NATIVE_INT64 >= 0 && a->NATIVE_INT32 >= 0 && a->P_LINK_CREATE >= 0;
    } else
      ok = false;
    if (!ok) {
      tried += c + " (unusable); ";
      dlclose(h);
      continue;
    }
    a->Eset_auto2(0, nullptr, nullptr);   // errors are reported through wf_last_error, not printed
    g_h5 = a.release();
    return g_h5;
  }
  wf::set_error("HDF5 library not available: " + tried);
  return nullptr;
}

std::string dirname_of(const std::string& p)
{
  const size_t s = p.find_last_of('/');
  return s == std::string::npos ? std::string(".") : p.substr(0, s);
}

// "<DataItem ...>file.h5:/path</DataItem>" of the first DataItem inside the first <tag ...> after `from`
bool data_item(const std::string& xml, size_t from, size_t to, const char* tag, std::string* file, std::string* path)
{
  const size_t t = xml.find(std::string("<") + tag, from);
  if (t == std::string::npos || t >= to) return false;
  const size_t d = xml.find("<DataItem", t);
  if (d == std::string::npos || d >= to) return false;
  const size_t b = xml.find('>', d), e = xml.find("</DataItem>", d);
  if (b == std::string::npos || e == std::string::npos) return false;
  std::string body = xml.substr(b + 1, e - b - 1);
  const size_t f = body.find_first_not_of(" \t\r\n"), l = body.find_last_not_of(" \t\r\n");
  if (f == std::string::npos) return false;
  body = body.substr(f, l - f + 1);
  const size_t c = body.find(":/");
  if (c == std::string::npos) return false;
  *file = body.substr(0, c);
  *path = body.substr(c + 1);
  return true;
}

bool grid_block(const std::string& xml, const char* name, size_t* from, size_t* to)
{
  const std::string key = std::string("Name=\"") + name + "\"";
  size_t g = 0;
  for (;;) {
    g = xml.find("<Grid", g);
    if (g == std::string::npos) return false;
    const size_t close = xml.find('>', g);
    if (close != std::string::npos && xml.substr(g, close - g).find(key) != std::string::npos) break;
    ++g;
  }
  *from = g;
  const size_t e = xml.find("</Grid>", g);
  *to = e == std::string::npos ? xml.size() : e;
  return true;
}

struct Dataset {
  std::vector<hsize_t> dims;
  hid_t id = -1;
};

}  // namespace

struct wf_mesh_file {
  std::string xml, dir;
  hid_t file = -1;
  std::string h5name;      // file currently open
  std::string meshfile;    // file of the mesh grid's topology / geometry
  std::string topo, geom;
};

namespace {

int open_h5(wf_mesh_file* m, const std::string& name)
{
  H5* a = h5();
  if (!a) return WF_ERR_INVALID;
  if (m->file >= 0 && name == m->h5name) return WF_OK;
  if (m->file >= 0) a->Fclose(m->file);
  const std::string full = name.empty() || name[0] == '/' ? name : m->dir + "/" + name;
  m->file = a->Fopen(full.c_str(), 0u /* H5F_ACC_RDONLY */, 0 /* H5P_DEFAULT */);
  if (m->file < 0) {
    wf::set_error("wf_mesh: cannot open " + full);
    return WF_ERR_INVALID;
  }
  m->h5name = name;
  return WF_OK;
}

int dataset_dims(wf_mesh_file* m, const std::string& path, std::vector<hsize_t>* dims)
{
  H5* a = h5();
  const hid_t d = a->Dopen2(m->file, path.c_str(), 0);
  if (d < 0) {
    wf::set_error("wf_mesh: no dataset " + path);
    return WF_ERR_INVALID;
  }
  const hid_t s = a->Dget_space(d);
  const int nd = a->Sget_simple_extent_ndims(s);
  dims->assign(nd > 0 ? nd : 0, 0);
  if (nd > 0) a->Sget_simple_extent_dims(s, dims->data(), nullptr);
  a->Sclose(s);
  a->Dclose(d);
  return WF_OK;
}

template <typename T>
int dataset_read(wf_mesh_file* m, const std::string& path, hid_t type, T* out)
{
  H5* a = h5();
  const hid_t d = a->Dopen2(m->file, path.c_str(), 0);
  if (d < 0) {
    wf::set_error("wf_mesh: no dataset " + path);
    return WF_ERR_INVALID;
  }
  const herr_t rc = a->Dread(d, type, 0 /* H5S_ALL */, 0, 0, out);
  a->Dclose(d);
  if (rc < 0) {
    wf::set_error("wf_mesh: reading " + path + " failed");
    return WF_ERR_INVALID;
  }
  return WF_OK;
}

}  // namespace

extern "C" {

int wf_mesh_open(const char* xdmf_path, const char* grid_name, wf_mesh_file** out)
{
  WF_REQUIRE(xdmf_path && grid_name && out, "wf_mesh_open: null argument");
  *out = nullptr;
  std::ifstream f(xdmf_path);
  if (!f) {
    wf::set_error(std::string("wf_mesh_open: cannot open ") + xdmf_path);
    return WF_ERR_INVALID;
  }
  std::stringstream ss;
  ss << f.rdbuf();
  auto m = std::make_unique<wf_mesh_file>();
  m->xml = ss.str();
  m->dir = dirname_of(xdmf_path);
  size_t from, to;
  if (!grid_block(m->xml, grid_name, &from, &to)) {
    wf::set_error(std::string("wf_mesh_open: no <Grid Name=\"") + grid_name + "\">");
    return WF_ERR_INVALID;
  }
  std::string file, file2;
  if (!data_item(m->xml, from, to, "Topology", &file, &m->topo) || !data_item(m->xml, from, to, "Geometry", &file2, &m->geom)) {
    wf::set_error("wf_mesh_open: the grid needs <Topology> and <Geometry> with HDF DataItems");
    return WF_ERR_INVALID;
  }
  // one .h5 per handle: a grid whose geometry lives in another file than its topology is refused, not
  // silently read from the wrong file
  if (file2 != file) {
    wf::set_error("wf_mesh_open: <Topology> and <Geometry> refer to different HDF5 files (" + file + ", " + file2 + ")");
    return WF_ERR_UNSUPPORTED;
  }
  m->meshfile = file;
  int rc = open_h5(m.get(), file);
  if (rc != WF_OK) return rc;
  *out = m.release();
  return WF_OK;
}

int wf_mesh_sizes(wf_mesh_file* m, int64_t* nverts, int64_t* ncells)
{
  WF_REQUIRE(m && nverts && ncells, "wf_mesh_sizes: null argument");
  std::vector<hsize_t> dt, dg;
  int rc;
  if ((rc = open_h5(m, m->meshfile)) != WF_OK) return rc;   // the tag grids may live in another file
  if ((rc = dataset_dims(m, m->topo, &dt)) != WF_OK || (rc = dataset_dims(m, m->geom, &dg)) != WF_OK) return rc;
  if (dt.size() != 2 || dt[1] != 8) {
    wf::set_error("wf_mesh_sizes: topology must be [ncells][8] (hexahedra)");   // the operators are hexahedral
    return WF_ERR_UNSUPPORTED;
  }
  if (dg.size() != 2 || (dg[1] != 3 && dg[1] != 2)) {
    wf::set_error("wf_mesh_sizes: geometry must be [nverts][3]");
    return WF_ERR_UNSUPPORTED;
  }
  *ncells = (int64_t)dt[0];
  *nverts = (int64_t)dg[0];
  return WF_OK;
}

int wf_mesh_read(wf_mesh_file* m, double* h_xverts, int32_t* h_cells)
{
  WF_REQUIRE(m && h_xverts && h_cells, "wf_mesh_read: null argument");
  int64_t nv, nc;
  int rc = wf_mesh_sizes(m, &nv, &nc);
  if (rc != WF_OK) return rc;
  std::vector<hsize_t> dg;
  if ((rc = dataset_dims(m, m->geom, &dg)) != WF_OK) return rc;   // wf_mesh_sizes has (re)opened the mesh file
  std::vector<double> g((size_t)nv * dg[1]);
  if ((rc = dataset_read(m, m->geom, h5()->NATIVE_DOUBLE, g.data())) != WF_OK) return rc;
  for (int64_t v = 0; v < nv; ++v)
    for (int d = 0; d < 3; ++d) h_xverts[v * 3 + d] = d < (int)dg[1] ? g[v * dg[1] + d] : 0.0;
  std::vector<int64_t> t((size_t)nc * 8);
  if ((rc = dataset_read(m, m->topo, h5()->NATIVE_INT64, t.data())) != WF_OK) return rc;
  static const int kTensorFromXdmf[8] = {0, 1, 3, 2, 4, 5, 7, 6};
  for (int64_t c = 0; c < nc; ++c)
    for (int v = 0; v < 8; ++v) {
      const int64_t id = t[c * 8 + kTensorFromXdmf[v]];
      WF_REQUIRE(id >= 0 && id < nv, "wf_mesh_read: vertex index out of range");
      h_cells[c * 8 + v] = (int32_t)id;
    }
  return WF_OK;
}

int wf_mesh_tags_size(wf_mesh_file* m, const char* tags_name, int64_t* nfacets)
{
  WF_REQUIRE(m && tags_name && nfacets, "wf_mesh_tags_size: null argument");
  size_t from, to;
  if (!grid_block(m->xml, tags_name, &from, &to)) {
    wf::set_error(std::string("wf_mesh_tags_size: no <Grid Name=\"") + tags_name + "\">");
    return WF_ERR_INVALID;
  }
  std::string file, topo;
  if (!data_item(m->xml, from, to, "Topology", &file, &topo)) {
    wf::set_error("wf_mesh_tags_size: the tag grid needs a <Topology> DataItem");
    return WF_ERR_INVALID;
  }
  int rc = open_h5(m, file);
  if (rc != WF_OK) return rc;
  std::vector<hsize_t> dt;
  if ((rc = dataset_dims(m, topo, &dt)) != WF_OK) return rc;
  if (dt.size() != 2 || dt[1] != 4) {
    wf::set_error("wf_mesh_tags_size: tag topology must be [nfacets][4] (quadrilaterals)");
    return WF_ERR_UNSUPPORTED;
  }
  *nfacets = (int64_t)dt[0];
  return WF_OK;
}

int wf_mesh_read_tags(wf_mesh_file* m, const char* tags_name, int32_t* h_facet_verts, int32_t* h_values)
{
  WF_REQUIRE(m && tags_name && h_facet_verts && h_values, "wf_mesh_read_tags: null argument");
  int64_t nf, nv = 0, nc = 0;
  int rc = wf_mesh_sizes(m, &nv, &nc);
  if (rc != WF_OK) return rc;
  if ((rc = wf_mesh_tags_size(m, tags_name, &nf)) != WF_OK) return rc;
  size_t from, to;
  grid_block(m->xml, tags_name, &from, &to);
  std::string file_t, file_v, topo, vals;
  data_item(m->xml, from, to, "Topology", &file_t, &topo);
  if (!data_item(m->xml, from, to, "Attribute", &file_v, &vals)) {
    wf::set_error("wf_mesh_read_tags: the tag grid needs an <Attribute> DataItem");
    return WF_ERR_INVALID;
  }
  if ((rc = open_h5(m, file_t)) != WF_OK) return rc;
  std::vector<int64_t> t((size_t)nf * 4);
  if ((rc = dataset_read(m, topo, h5()->NATIVE_INT64, t.data())) != WF_OK) return rc;
  static const int kTensorFromXdmf[4] = {0, 1, 3, 2};
  for (int64_t f = 0; f < nf; ++f)
    for (int v = 0; v < 4; ++v) {
      const int64_t id = t[f * 4 + kTensorFromXdmf[v]];
      WF_REQUIRE(id >= 0 && id < nv, "wf_mesh_read_tags: facet vertex index out of range");
      h_facet_verts[f * 4 + v] = (int32_t)id;
    }
  // the caller's h_values has nf entries (the size wf_mesh_tags_size reported from the tag TOPOLOGY): the
  // Values dataset -- in the file ITS DataItem names -- must have exactly that many, as [nf] or [nf][1]
  if ((rc = open_h5(m, file_v)) != WF_OK) return rc;
  std::vector<hsize_t> dv;
  if ((rc = dataset_dims(m, vals, &dv)) != WF_OK) return rc;
  hsize_t nvals = 1;
  for (hsize_t d : dv) nvals *= d;
  if (dv.empty() || dv.size() > 2 || (dv.size() == 2 && dv[1] != 1) || nvals != (hsize_t)nf) {
    wf::set_error("wf_mesh_read_tags: the Values dataset must hold one value per tagged facet ([nfacets] or [nfacets][1])");
    return WF_ERR_INVALID;
  }
  return dataset_read(m, vals, h5()->NATIVE_INT32, h_values);
}

int wf_mesh_close(wf_mesh_file* m)
{
  if (!m) return WF_OK;
  if (m->file >= 0 && g_h5) g_h5->Fclose(m->file);
  delete m;
  return WF_OK;
}

// Writer (XDMF + HDF5 in the layout above; cells / facets given in TENSOR vertex order).
int wf_mesh_write(const char* xdmf_path, const char* grid_name, int64_t nverts, const double* h_xverts, int64_t ncells,
                  const int32_t* h_cells, const char* tags_name, int64_t nfacets, const int32_t* h_facet_verts,
                  const int32_t* h_values)
{
  WF_REQUIRE(xdmf_path && grid_name && h_xverts && h_cells && nverts >= 0 && ncells >= 0, "wf_mesh_write: bad arguments");
  WF_REQUIRE(!tags_name || nfacets == 0 || (h_facet_verts && h_values), "wf_mesh_write: tag arrays missing");
  H5* a = h5();
  if (!a) return WF_ERR_INVALID;
  std::string xp = xdmf_path, hp = xp;
  const size_t dot = hp.rfind(".xdmf");
  hp = (dot == std::string::npos ? hp : hp.substr(0, dot)) + ".h5";
  const std::string hname = hp.substr(hp.find_last_of('/') == std::string::npos ? 0 : hp.find_last_of('/') + 1);
  const hid_t file = a->Fcreate(hp.c_str(), 2u /* H5F_ACC_TRUNC */, 0, 0);
  if (file < 0) {
    wf::set_error("wf_mesh_write: cannot create " + hp);
    return WF_ERR_INVALID;
  }
  const hid_t lcpl = a->Pcreate(a->P_LINK_CREATE);
  a->Pset_create_intermediate_group(lcpl, 1);
  auto put = [&](const std::string& path, hid_t type, int rank, const hsize_t* dims, const void* data) -> bool {
    const hid_t sp = a->Screate_simple(rank, dims, nullptr);
    const hid_t d = a->Dcreate2(file, path.c_str(), type, sp, lcpl, 0, 0);
    const bool ok = d >= 0 && a->Dwrite(d, type, 0, 0, 0, data) >= 0;
    if (d >= 0) a->Dclose(d);
    a->Sclose(sp);
    return ok;
  };
  static const int kXdmfFromTensor8[8] = {0, 1, 3, 2, 4, 5, 7, 6}, kXdmfFromTensor4[4] = {0, 1, 3, 2};   // involutions
  std::vector<int64_t> t((size_t)ncells * 8);
  for (int64_t c = 0; c < ncells; ++c)
    for (int v = 0; v < 8; ++v) t[c * 8 + v] = h_cells[c * 8 + kXdmfFromTensor8[v]];
  const std::string gname = grid_name;
  const hsize_t dt[2] = {(hsize_t)ncells, 8}, dg[2] = {(hsize_t)nverts, 3};
  bool ok = put("/Mesh/" + gname + "/topology", a->NATIVE_INT64, 2, dt, t.data())
            && put("/Mesh/" + gname + "/geometry", a->NATIVE_DOUBLE, 2, dg, h_xverts);
  std::ostringstream x;
  x << "<?xml version=\"1.0\"?>\n<Xdmf Version=\"3.0\" xmlns:xi=\"http://www.w3.org/2001/XInclude\">\n  <Domain>\n"
    << "    <Grid Name=\"" << gname << "\" GridType=\"Uniform\">\n"
    << "      <Topology TopologyType=\"Hexahedron\" NumberOfElements=\"" << ncells << "\" NodesPerElement=\"8\">\n"
    << "        <DataItem Dimensions=\"" << ncells << " 8\" NumberType=\"Int\" Format=\"HDF\">" << hname << ":/Mesh/" << gname
    << "/topology</DataItem>\n      </Topology>\n      <Geometry GeometryType=\"XYZ\">\n"
    << "        <DataItem Dimensions=\"" << nverts << " 3\" Format=\"HDF\">" << hname << ":/Mesh/" << gname
    << "/geometry</DataItem>\n      </Geometry>\n    </Grid>\n";
  if (ok && tags_name) {
    const std::string tn = tags_name;
    std::vector<int64_t> ft((size_t)nfacets * 4);
    for (int64_t f = 0; f < nfacets; ++f)
      for (int v = 0; v < 4; ++v) ft[f * 4 + v] = h_facet_verts[f * 4 + kXdmfFromTensor4[v]];
    const hsize_t d4[2] = {(hsize_t)nfacets, 4}, d1[1] = {(hsize_t)nfacets};
    ok = put("/MeshTags/" + tn + "/topology", a->NATIVE_INT64, 2, d4, ft.data())
         && put("/MeshTags/" + tn + "/Values", a->NATIVE_INT32, 1, d1, h_values);
    x << "    <Grid Name=\"" << tn << "\" GridType=\"Uniform\">\n"
      << "      <xi:include xpointer=\"xpointer(/Xdmf/Domain/Grid/Geometry)\" />\n"
      << "      <Topology TopologyType=\"Quadrilateral\" NumberOfElements=\"" << nfacets << "\" NodesPerElement=\"4\">\n"
      << "        <DataItem Dimensions=\"" << nfacets << " 4\" NumberType=\"Int\" Format=\"HDF\">" << hname << ":/MeshTags/" << tn
      << "/topology</DataItem>\n      </Topology>\n"
      << "      <Attribute Name=\"" << tn << "\" AttributeType=\"Scalar\" Center=\"Cell\">\n"
      << "        <DataItem Dimensions=\"" << nfacets << " 1\" Format=\"HDF\">" << hname << ":/MeshTags/" << tn
      << "/Values</DataItem>\n      </Attribute>\n    </Grid>\n";
  }
  x << "  </Domain>\n</Xdmf>\n";
  a->Pclose(lcpl);
  a->Fclose(file);
  if (!ok) {
    wf::set_error("wf_mesh_write: writing " + hp + " failed");
    return WF_ERR_INVALID;
  }
  std::ofstream xf(xdmf_path);
  xf << x.str();
  if (!xf) {
    wf::set_error(std::string("wf_mesh_write: cannot write ") + xdmf_path);
    return WF_ERR_INVALID;
  }
  return WF_OK;
}

}  // extern "C"
