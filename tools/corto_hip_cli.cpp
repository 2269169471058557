// corto_hip — the `corto` command line tool (upstream src/main.cpp:48-388) on this repo's encoder and GPU decoder.
//
//   corto_hip [OPTIONS] <FILE.ply>
//
// Same options, same defaults and the same flow as upstream: load the model, encode it to <output>.crt, and - with
// -P <file.ply> - decode the blob again (here: on the MI355X through the crt::Decoder facade) and save the decoded mesh
// as a binary PLY laid out like upstream's MeshLoader::savePly (src/meshloader.cpp:294-329), so both files can be compared
// byte for byte with upstream's (tests/test_cli_*.py).  Differences: upstream always runs the round-trip decode to print
// its speed, this tool only when -P asks for the result (the decode needs a GPU; there is no CPU fallback); .obj input,
// per-wedge texture coordinates and `texnumber` groups (upstream's tinyply/objload loaders) are not read.
//
// Build: g++ -O2 -std=c++17 -I include tools/corto_hip_cli.cpp -L corto_amd/lib -lcorto_hip   (python -m corto_amd.build)
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "corto/decoder.h"
#include "corto_hip.h"

namespace {

struct Model {                                  // what upstream's MeshLoader holds after loadPly (src/meshloader.h:45-57)
	uint32_t nvert = 0, nface = 0, ncolor = 0;
	std::vector<float> coords, norms, uvs, radiuses;
	std::vector<uint8_t> colors;
	std::vector<uint32_t> index;
};

// ---- PLY in: ascii / binary_little_endian, scalar properties of any PLY type, one list property per face ----------------
struct Prop { std::string name; int type = 0, list_count_type = -1; };      // type: index into kTypes
const char *const kTypes[][2] = {{"char", "int8"}, {"uchar", "uint8"}, {"short", "int16"}, {"ushort", "uint16"},
                                 {"int", "int32"}, {"uint", "uint32"}, {"float", "float32"}, {"double", "float64"}};
const int kSize[] = {1, 1, 2, 2, 4, 4, 4, 8};

int type_of(const std::string &t) {
	for(int i = 0; i < 8; i++) if(t == kTypes[i][0] || t == kTypes[i][1]) return i;
	return -1;
}

struct Reader {
	std::istream &in;
	bool ascii;
	double scalar(int type) {
		if(ascii) { double v = 0; in >> v; return v; }
		unsigned char b[8] = {0};
		in.read((char *)b, kSize[type]);
		switch(type) {
		case 0: return (double)(int8_t)b[0];
		case 1: return (double)b[0];
		case 2: { int16_t v; memcpy(&v, b, 2); return v; }
		case 3: { uint16_t v; memcpy(&v, b, 2); return v; }
		case 4: { int32_t v; memcpy(&v, b, 4); return v; }
		case 5: { uint32_t v; memcpy(&v, b, 4); return v; }
		case 6: { float v; memcpy(&v, b, 4); return v; }
		default: { double v; memcpy(&v, b, 8); return v; }
		}
	}
};

bool load_ply(const std::string &path, Model &m, std::string &err) {
	std::ifstream in(path, std::ios::binary);
	if(!in.is_open()) { err = "cannot open " + path; return false; }
	std::string line;
	std::getline(in, line);
	if(line.substr(0, 3) != "ply") { err = "not a PLY file"; return false; }
	struct Element { std::string name; size_t count = 0; std::vector<Prop> props; };
	std::vector<Element> elements;
	bool ascii = false, have_format = false;
	while(std::getline(in, line)) {
		if(!line.empty() && line.back() == '\r') line.pop_back();
		std::istringstream ls(line);
		std::string tok;
		ls >> tok;
		if(tok == "end_header") break;
		if(tok == "format") {
			std::string f; ls >> f;
			if(f == "ascii") ascii = true;
			else if(f != "binary_little_endian") { err = "unsupported PLY format " + f; return false; }
			have_format = true;
		} else if(tok == "element") {
			Element e; ls >> e.name >> e.count; elements.push_back(e);
		} else if(tok == "property" && !elements.empty()) {
			Prop p; std::string t; ls >> t;
			if(t == "list") { std::string ct, it; ls >> ct >> it >> p.name; p.list_count_type = type_of(ct); p.type = type_of(it); }
			else { p.type = type_of(t); ls >> p.name; }
			if(p.type < 0) { err = "unknown PLY property type in: " + line; return false; }
			elements.back().props.push_back(p);
		}
	}
	if(!have_format) { err = "PLY header without format"; return false; }
	Reader R{in, ascii};
	for(const Element &e : elements) {
		if(e.name == "vertex") {
			// columns upstream asks tinyply for (src/meshloader.cpp:52-64): the first uv naming that is present wins
			std::map<std::string, int> col;
			for(size_t k = 0; k < e.props.size(); k++) col[e.props[k].name] = (int)k;
			auto has = [&](std::initializer_list<const char *> names) { for(auto n : names) if(!col.count(n)) return false; return true; };
			const bool P = has({"x", "y", "z"}), N = has({"nx", "ny", "nz"}), C3 = has({"red", "green", "blue"}), C4 = C3 && col.count("alpha");
			const char *un = nullptr, *vn = nullptr;
			if(has({"texture_u", "texture_v"})) { un = "texture_u"; vn = "texture_v"; }
			else if(has({"s", "t"})) { un = "s"; vn = "t"; }
			else if(has({"u", "v"})) { un = "u"; vn = "v"; }
			const bool Rd = col.count("radius") != 0;
			if(!P) { err = "PLY vertex element without x, y, z"; return false; }
			m.nvert = (uint32_t)e.count;
			m.coords.resize(e.count*3);
			if(N) m.norms.resize(e.count*3);
			m.ncolor = C4 ? 4 : 0;                 // upstream requests red, green, blue, alpha together (4 components or none)
			if(C4) m.colors.resize(e.count*4);
			if(un) m.uvs.resize(e.count*2);
			if(Rd) m.radiuses.resize(e.count);
			std::vector<double> row(e.props.size());
			for(size_t i = 0; i < e.count; i++) {
				for(size_t k = 0; k < e.props.size(); k++) {
					if(e.props[k].list_count_type >= 0) { const size_t n = (size_t)R.scalar(e.props[k].list_count_type); for(size_t j = 0; j < n; j++) R.scalar(e.props[k].type); row[k] = 0; }
					else row[k] = R.scalar(e.props[k].type);
				}
				m.coords[i*3] = (float)row[col["x"]]; m.coords[i*3 + 1] = (float)row[col["y"]]; m.coords[i*3 + 2] = (float)row[col["z"]];
				if(N) { m.norms[i*3] = (float)row[col["nx"]]; m.norms[i*3 + 1] = (float)row[col["ny"]]; m.norms[i*3 + 2] = (float)row[col["nz"]]; }
				if(C4) { m.colors[i*4] = (uint8_t)row[col["red"]]; m.colors[i*4 + 1] = (uint8_t)row[col["green"]]; m.colors[i*4 + 2] = (uint8_t)row[col["blue"]]; m.colors[i*4 + 3] = (uint8_t)row[col["alpha"]]; }
				if(un) { m.uvs[i*2] = (float)row[col[un]]; m.uvs[i*2 + 1] = (float)row[col[vn]]; }
				if(Rd) m.radiuses[i] = (float)row[col["radius"]];
			}
		} else {
			const bool face = e.name == "face";
			if(face) m.index.reserve(e.count*3);
			for(size_t i = 0; i < e.count; i++)
				for(const Prop &p : e.props) {
					if(p.list_count_type < 0) { R.scalar(p.type); continue; }
					const size_t n = (size_t)R.scalar(p.list_count_type);
					const bool idx = face && (p.name == "vertex_index" || p.name == "vertex_indices");
					if(idx && n != 3) { err = "only triangles are supported"; return false; }
					for(size_t j = 0; j < n; j++) { const double v = R.scalar(p.type); if(idx) m.index.push_back((uint32_t)v); }
				}
		}
		if(!in && !in.eof()) { err = "PLY file ends early"; return false; }
	}
	m.nface = (uint32_t)(m.index.size()/3);
	for(uint32_t v : m.index) if(v >= m.nvert) { err = "face index out of range"; return false; }
	return true;
}

// ---- PLY out: the layout of upstream's savePly through tinyply's binary writer ------------------------------------------
bool save_ply(const std::string &path, const Model &m) {
	FILE *f = fopen(path.c_str(), "wb");
	if(!f) return false;
	std::string h = "ply\nformat binary_little_endian 1.0\n";
	h += "element vertex " + std::to_string(m.nvert) + "\nproperty float x\nproperty float y\nproperty float z\n";
	if(!m.norms.empty()) h += "property float nx\nproperty float ny\nproperty float nz\n";
	if(!m.colors.empty()) { h += "property uchar red\nproperty uchar green\nproperty uchar blue\n"; if(m.ncolor == 4) h += "property uchar alpha\n"; }
	if(!m.uvs.empty()) h += "property float texture_u\nproperty float texture_v\n";
	if(!m.radiuses.empty()) h += "property float radius\n";
	if(m.nface) h += "element face " + std::to_string(m.nface) + "\nproperty list uchar uint vertex_indices\n";
	h += "end_header\n";
	fwrite(h.data(), 1, h.size(), f);
	for(uint32_t i = 0; i < m.nvert; i++) {
		fwrite(&m.coords[(size_t)i*3], 4, 3, f);
		if(!m.norms.empty()) fwrite(&m.norms[(size_t)i*3], 4, 3, f);
		if(!m.colors.empty()) fwrite(&m.colors[(size_t)i*m.ncolor], 1, m.ncolor, f);
		if(!m.uvs.empty()) fwrite(&m.uvs[(size_t)i*2], 4, 2, f);
		if(!m.radiuses.empty()) fwrite(&m.radiuses[i], 4, 1, f);
	}
	for(uint32_t i = 0; i < m.nface; i++) { const uint8_t three = 3; fwrite(&three, 1, 1, f); fwrite(&m.index[(size_t)i*3], 4, 3, f); }
	return fclose(f) == 0;
}

void usage() {
	std::cerr <<
R"use(Usage: corto_hip [OPTIONS] <FILE>

FILE is the path to a .ply 3D model.
  -o <output>: filename of the .crt compressed file.
               if not specified the extension of the input file will be replaced.
  -e <key=value>: add an exif property, or more than one.
  -p : treat the input as a point cloud.
  -v <bits>: vertex bits quantization. If not specified an euristic is used
  -n <bits>: normal bits quantization. Default 10.
  -c <bits>: color bits quantization. Default 6.
  -u <bits>: texture coordinate bits. Default 12.
  -q <step>: quantization step unit (float) instead of bits for vertex coordinates
  -N <prediction>: normal prediction can be: delta, estimated, border (default)
  -P <file.ply>: decompress on the GPU and save as .ply
)use";
}

bool ends_with(const std::string &s, const std::string &x) { return s.size() >= x.size() && !s.compare(s.size() - x.size(), x.size(), x); }

} // namespace

int main(int argc, char *argv[]) {
	std::string input, output, plyfile, normal_prediction;
	bool pointcloud = false;
	float vertex_q = 0.0f;
	int vertex_bits = 0, norm_bits = 10, r_bits = 6, g_bits = 7, b_bits = 6, a_bits = 5, uv_bits = 12;     // src/main.cpp:81-88
	std::map<std::string, std::string> exif;
	int c;
	while((c = getopt(argc, argv, "pAo:v:n:c:u:q:N:e:P:G:")) != -1) {
		switch(c) {
		case 'o': output = optarg; break;
		case 'p': pointcloud = true; break;
		case 'v': vertex_bits = atoi(optarg); break;
		case 'n': norm_bits = atoi(optarg); break;
		case 'c': r_bits = g_bits = a_bits = b_bits = atoi(optarg); break;
		case 'u': uv_bits = atoi(optarg); break;
		case 'q': vertex_q = (float)atof(optarg); break;
		case 'N': normal_prediction = optarg; break;
		case 'P': plyfile = optarg; break;
		case 'A': std::cerr << "-A (add normals) is not supported" << std::endl; return 1;
		case 'G': std::cerr << "-G applies to .obj input, which is not supported" << std::endl; return 1;
		case 'e': {
			const std::string opt(optarg);
			const size_t pos = opt.find('=');
			if(pos == std::string::npos || pos == 0 || pos == opt.size() - 1) { std::cerr << "Expecting key=value or \"key=another value\" for exif arguments" << std::endl; return 1; }
			exif[opt.substr(0, pos)] = opt.substr(pos + 1);
			break;
		}
		case '?': usage(); return 0;
		default: usage(); return 1;
		}
	}
	if(optind == argc) { std::cerr << "Missing filename" << std::endl; usage(); return 1; }
	if(optind != argc - 1) { std::cerr << "Too many arguments\n"; usage(); return 1; }
	input = argv[optind];
	if(!ends_with(input, ".ply") && !ends_with(input, ".PLY")) { std::cerr << "Failed loading model: " << input << " (only .ply input is supported)" << std::endl; return 1; }

	Model in;
	std::string err;
	if(!load_ply(input, in, err)) { std::cerr << "Failed loading model: " << input << " (" << err << ")" << std::endl; return 1; }
	const uint32_t group_end = (uint32_t)(in.index.size()/3);            // loadPly: one group holding every face (src/meshloader.cpp:121), kept under -p
	if(pointcloud) in.nface = 0;
	pointcloud = in.nface == 0;
	int prediction = 2;                                                   // BORDER, src/main.cpp:163
	if(!normal_prediction.empty()) {
		if(normal_prediction == "delta") prediction = 0;
		else if(normal_prediction == "border") prediction = 2;
		else if(normal_prediction == "estimated") prediction = 1;
		else { std::cerr << "Unknown normal prediction: " << normal_prediction << " expecting: delta, border or estimated" << std::endl; return 1; }
	}

	// ---- encode (src/main.cpp:180-222) ----
	crthip_mesh M;
	memset(&M, 0, sizeof(M));
	M.nvert = in.nvert; M.nface = in.nface;
	M.position = in.coords.data();
	M.index = pointcloud ? nullptr : in.index.data();
	M.position_bits = vertex_bits; M.position_q = vertex_q;               // both 0: upstream's heuristic step
	if(!in.norms.empty() && norm_bits > 0) { M.normal = in.norms.data(); M.normal_bits = norm_bits; M.normal_prediction = prediction; }
	if(!in.colors.empty() && r_bits > 0) {
		M.color = in.colors.data(); M.color_components = (int32_t)in.ncolor;
		M.color_bits[0] = r_bits; M.color_bits[1] = g_bits; M.color_bits[2] = b_bits; M.color_bits[3] = a_bits;
	}
	if(!in.uvs.empty() && uv_bits > 0) { M.uv = in.uvs.data(); M.uv_q = (float)pow(2, -uv_bits); }
	if(!in.radiuses.empty()) { M.radius = in.radiuses.data(); M.radius_q = 1.0f; }
	M.group_end = &group_end; M.ngroups = 1;
	M.entropy = CRTHIP_ENTROPY_TUNSTALL;
	std::string ex;
	for(auto &kv : exif) { ex += kv.first; ex.push_back('\0'); ex += kv.second; ex.push_back('\0'); }
	M.exif = ex.data(); M.nexif = (uint32_t)exif.size();
	uint32_t nvert = 0, nface = 0;
	const int64_t size = crthip_encode(&M, nullptr, 0, &nvert, &nface);
	if(size < 0) { std::cerr << "Encoding failed: " << crthip_last_error() << std::endl; return 1; }
	// the blob must sit on a 4-byte boundary for the decoder (src/decoder.cpp:43-44)
	std::vector<uint32_t> blob32(((size_t)size + 3)/4 + 1);
	uint8_t *blob = (uint8_t *)blob32.data();
	if(crthip_encode(&M, blob, (size_t)size, &nvert, &nface) != size) { std::cerr << "Encoding failed: " << crthip_last_error() << std::endl; return 1; }
	std::cout << "Nvert: " << nvert << " Nface: " << nface << std::endl;
	std::cout << "Compressed to: " << size << std::endl;
	std::cout << "Ratio: " << 100.0f*size/(nvert*12 + nface*12) << "%" << std::endl;
	std::cout << "Bpv: " << 8.0f*size/nvert << std::endl << std::endl;

	// ---- decode on the GPU and save, when asked to (src/main.cpp:266-300, 330-331) ----
	if(!plyfile.empty()) {
		Model out;
		try {
			crt::Decoder decoder((int)size, blob);
			if(decoder.nface != nface || decoder.nvert != nvert) { std::cerr << "Decoder disagrees with the encoder about the mesh size" << std::endl; return 1; }
			out.nvert = nvert; out.nface = nface;
			out.coords.resize((size_t)nvert*3);
			decoder.setPositions(out.coords.data());
			if(decoder.data.count("normal")) { out.norms.resize((size_t)nvert*3); decoder.setNormals(out.norms.data()); }
			if(decoder.data.count("color")) { out.ncolor = in.ncolor; out.colors.resize((size_t)nvert*in.ncolor); decoder.setColors(out.colors.data(), (int)in.ncolor); }
			if(decoder.data.count("uv")) { out.uvs.resize((size_t)nvert*2); decoder.setUvs(out.uvs.data()); }
			if(decoder.data.count("radius")) { out.radiuses.resize(nvert); decoder.setAttribute("radius", (char *)out.radiuses.data(), crt::VertexAttribute::FLOAT); }
			if(decoder.nface) { out.index.resize((size_t)nface*3); decoder.setIndex(out.index.data()); }
			decoder.decode();
		} catch(const char *e) { std::cerr << "Decoding failed: " << e << std::endl; return 1; }
		if(!save_ply(plyfile, out)) { std::cerr << "Failed saving file: " << plyfile << std::endl; return 1; }
	}

	if(output.empty()) output = input.substr(0, input.find_last_of("."));
	if(!ends_with(output, ".crt")) output += ".crt";
	FILE *file = fopen(output.c_str(), "wb");
	if(!file) { std::cerr << "Could not open file: " << output << std::endl; return 1; }
	const size_t written = fwrite(blob, 1, (size_t)size, file);
	fclose(file);
	if(written != (size_t)size) { std::cerr << "Failed saving file: " << output << std::endl; return 1; }
	return 0;
}
