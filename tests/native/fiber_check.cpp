#define ZKFIBER_IMPLEMENTATION
#include "fiber.hpp"
#include <cstdio>
#include <vector>
#include <memory>
int main() {
    std::vector<std::unique_ptr<zkfiber::fiber>> fb;
    int order[32]; int n = 0;
    for (int k = 0; k < 3; ++k) fb.emplace_back(new zkfiber::fiber([k, &order, &n]() { for (int i = 0; i < 3; ++i) { order[n++] = 10 * k + i; zkfiber::fiber::yield(); } if (k == 1) throw std::runtime_error("x"); }));
    int caught = 0;
    for (bool any = true; any;) { any = false; for (auto &f : fb) if (!f->done()) { try { f->resume(); } catch (std::exception &) { ++caught; } any = any || !f->done(); } }
    for (int i = 0; i < n; ++i) printf("%d ", order[i]);
    printf("caught %d\n", caught);
    return 0;
}
