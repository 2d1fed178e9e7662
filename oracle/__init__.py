"""CPU oracle for the render path -- TEST INFRASTRUCTURE (see gsplat_oracle.cpp)."""
