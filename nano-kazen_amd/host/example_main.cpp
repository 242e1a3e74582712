// Example: the shape of nano-kazen's main.cpp (src/kazen/main.cpp:60-80) on top of the MI355X core — load a scene file,
// render it on GPU 0, write <scene>.png. Build:
//   g++ -std=c++17 -O2 example_main.cpp -L../csrc -lkazen_mi355x -Wl,-rpath,$PWD/../csrc -o kazen_mi355x
#include "kazen_sceneio.hpp"

#include <iostream>

int main(int argc, char **argv) {
    if (argc != 2) { std::cerr << "Syntax: " << argv[0] << " <scene.xml>" << std::endl; return -1; }
    try {
        std::unique_ptr<kazen::Object> root(kazen::loadFromXML(argv[1]));
        if (root->getClassType() != kazen::Object::EScene) throw kazen::Exception("The root element must be a scene");
        kazen::renderer::render(static_cast<kazen::Scene *>(root.get()), argv[1], 0);     // writes the PNG next to the scene file
    } catch (const std::exception &e) {
        std::cerr << "Fatal error: " << e.what() << std::endl;
        return -1;
    }
    return 0;
}
