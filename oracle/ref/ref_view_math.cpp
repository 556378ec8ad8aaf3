// oracle/ref/ref_view_math.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Pins the host matrix chain of ReconIntegration::draw() (framework/reconstruction/recon_integration.cpp:66-72 vol_to_world,
// :182-193 image_to_eye, :195-205 NormalMatrix / CameraPos) against the two matrix libraries the reference itself links:
// external/gloost/Matrix.cpp (compiled where it lies) and the vendored glm 0.9.5.3.  This file holds only the call sequence
// of those lines written against the libraries' public API -- no reference source is copied into the repository.
//
//   ref_view_math <32 floats: modelview, projection (column major)> <vw> <vh> <bbox min xyz> <bbox max xyz>
// prints, one per line, %.9g: vol_to_world[16], image_to_eye[16], NormalMatrix[16], CameraPos[3]
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <Matrix.h>                       // -I/root/reference/external/gloost
#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>
#include <glm/gtc/matrix_inverse.hpp>
#include <glm/gtc/type_ptr.hpp>

static void dump(const float* p, int n) { for (int i = 0; i < n; ++i) printf("%.9g\n", p[i]); }

int main(int argc, char** argv) {
  if (argc != 1 + 32 + 2 + 6) { fprintf(stderr, "usage: see header\n"); return 2; }
  float mv[16], pr[16];
  for (int i = 0; i < 16; ++i) { mv[i] = (float)atof(argv[1 + i]); pr[i] = (float)atof(argv[17 + i]); }
  const unsigned vw = (unsigned)atoi(argv[33]), vh = (unsigned)atoi(argv[34]);
  float bmin[3], bmax[3];
  for (int i = 0; i < 3; ++i) { bmin[i] = (float)atof(argv[35 + i]); bmax[i] = (float)atof(argv[38 + i]); }

  // :66-72
  glm::fvec3 dims{bmax[0] - bmin[0], bmax[1] - bmin[1], bmax[2] - bmin[2]};
  glm::fvec3 trans{bmin[0], bmin[1], bmin[2]};
  glm::fmat4 vol_to_world = glm::scale(glm::fmat4{1.0f}, dims);
  vol_to_world = glm::translate(glm::fmat4{1.0f}, trans) * vol_to_world;

  // :182-193 (gloost)
  gloost::Matrix projection; memcpy(projection.data(), pr, sizeof pr);
  gloost::Matrix vt; vt.setIdentity(); vt.setTranslate(1.0, 1.0, 1.0);
  gloost::Matrix vs; vs.setIdentity(); vs.setScale(vw * 0.5, vh * 0.5, 0.5f);
  gloost::Matrix image_to_eye = vs * vt * projection;
  image_to_eye.invert();

  // :195-205 (gloost -> glm)
  gloost::Matrix modelview; memcpy(modelview.data(), mv, sizeof mv);
  glm::fmat4 model_view{modelview};
  glm::fmat4 normal_matrix = glm::inverseTranspose(model_view * vol_to_world);
  glm::fvec4 camera_world{glm::inverse(model_view) * glm::fvec4{0.0f, 0.0f, 0.0f, 1.0f}};
  glm::vec3 camera_texturespace{glm::inverse(vol_to_world) * camera_world};

  dump(glm::value_ptr(vol_to_world), 16);
  dump(image_to_eye.data(), 16);
  dump(glm::value_ptr(normal_matrix), 16);
  dump(glm::value_ptr(camera_texturespace), 3);
  return 0;
}
