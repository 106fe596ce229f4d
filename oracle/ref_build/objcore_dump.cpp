// objcore_dump.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (container only).
//
// Driver around the REFERENCE's own OBJ loader, compiled from the sources where they lie
// (/root/reference/GPUPathTracer/GPUPathTracer/objcore/{objloader.cpp,obj.cpp} with the
// reference's vendored glm): runs objLoader on one file exactly as PathTracer::prepMesh does
// (path_tracer.cpp:52-56) and dumps what the loader holds afterwards -- points, normals, faces,
// face normals, bounding box -- as a flat little-endian binary that tools/make_goldens.py turns
// into tests/golden/obj_<name>.npz.  Nothing of the reference is copied: only its outputs leave.
//
//   objcore_dump <in.obj> <out.bin>
//
// Layout of out.bin (int32 / float32):  nPoints, nNormals, nFaces, nFaceNormals,
//   points[nPoints][3], normals[nNormals][3],
//   for each face: count, indices[count];  for each face-normal list: count, indices[count];
//   bbox min xyz, max xyz.
#include <cstdio>
#include <vector>

#include "objcore/objloader.h"

static void put_i(FILE* f, int v) { fwrite(&v, 4, 1, f); }
static void put_f(FILE* f, float v) { fwrite(&v, 4, 1, f); }

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: objcore_dump in.obj out.bin\n"); return 2; }
    obj* mesh = new obj();
    objLoader* loader = new objLoader(argv[1], mesh);      // path_tracer.cpp:53-54
    (void)loader;
    std::vector<glm::vec4>* points = mesh->getPoints();
    std::vector<glm::vec4>* normals = mesh->getNormals();
    std::vector<std::vector<int> >* faces = mesh->getFaces();
    std::vector<std::vector<int> >* facenormals = mesh->getFaceNormals();
    FILE* f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 1; }
    put_i(f, (int)points->size());
    put_i(f, (int)normals->size());
    put_i(f, (int)faces->size());
    put_i(f, (int)facenormals->size());
    for (size_t i = 0; i < points->size(); i++) { put_f(f, (*points)[i].x); put_f(f, (*points)[i].y); put_f(f, (*points)[i].z); }
    for (size_t i = 0; i < normals->size(); i++) { put_f(f, (*normals)[i].x); put_f(f, (*normals)[i].y); put_f(f, (*normals)[i].z); }
    for (size_t i = 0; i < faces->size(); i++) {
        put_i(f, (int)(*faces)[i].size());
        for (size_t k = 0; k < (*faces)[i].size(); k++) put_i(f, (*faces)[i][k]);
    }
    for (size_t i = 0; i < facenormals->size(); i++) {
        put_i(f, (int)(*facenormals)[i].size());
        for (size_t k = 0; k < (*facenormals)[i].size(); k++) put_i(f, (*facenormals)[i][k]);
    }
    glm::vec3 lo = mesh->getMin(), hi = mesh->getMax();
    put_f(f, lo.x); put_f(f, lo.y); put_f(f, lo.z);
    put_f(f, hi.x); put_f(f, hi.y); put_f(f, hi.z);
    fclose(f);
    return 0;
}
