// ORACLE (test infrastructure, not product code).  A driver around the REFERENCE's own header-only SHA-256, compiled from where it lies
// (/root/reference/src/utilities/sha256.h:39-94, the function that names engine files: img2img_build.cpp:8-27,151-155) - the one piece of the
// reference's hot-path neighbourhood that builds here without TensorRT / CUDA / OpenCV.  Reads lines from stdin, prints utils::sha256(line) per line.
// Built by oracle/ref/Makefile into oracle/_ref/sha256_ref (git-ignored; travels to the GPU box with the snapshot).  No reference source is copied.
#include <iostream>
#include <sstream>
#include <string>
#include "utilities/sha256.h"

int main() {
    std::string line;
    while (std::getline(std::cin, line)) std::cout << utils::sha256(line) << "\n";
    return 0;
}
