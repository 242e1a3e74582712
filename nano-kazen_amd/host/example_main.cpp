// Example: the shape of nano-kazen's main.cpp (src/kazen/main.cpp:60-80) on top of the MI355X core - load a scene file, render it on the
// GPUs of the node through the ADAPTER a maintainer adds to a kazen tree (adapter/renderer_mi355x.cpp, compiled unchanged), write <scene>.png.
// Build (from the repository root):
//   g++ -std=c++17 -O2 -I include -I nano-kazen_amd/host/mirror_tree -I nano-kazen_amd/host/adapter nano-kazen_amd/host/example_main.cpp \
//       nano-kazen_amd/host/adapter/renderer_mi355x.cpp -Lnano-kazen_amd/csrc -lkazen_mi355x -Wl,-rpath,$PWD/nano-kazen_amd/csrc -o kazen_mi355x
#include <kazen/renderer.h>
#include <kazen/scene.h>
#include "kazen_sceneio.hpp"

#include <iostream>

int main(int argc, char **argv) {
    if (argc != 2) { std::cerr << "Syntax: " << argv[0] << " <scene.xml>" << std::endl; return -1; }
    try {
        std::unique_ptr<kazen::Object> root(kazen::loadFromXML(argv[1]));
        if (root->getClassType() != kazen::Object::EScene) throw kazen::Exception("The root element must be a scene");
        kazen::renderer::render(static_cast<kazen::Scene *>(root.get()), argv[1]);        // writes the PNG next to the scene file (main.cpp:75)
    } catch (const std::exception &e) {
        std::cerr << "Fatal error: " << e.what() << std::endl;
        return -1;
    }
    return 0;
}
