// compile-hygiene stand-in (tests/adapter_stubs/README.md): the slice of g2o::SparseBlockMatrix the adapters use
// (object_slam/Thirdparty/g2o/g2o/core/sparse_block_matrix.h:57-104).  Declarations only.
#pragma once
namespace g2o {
template <class MatrixType>
class SparseBlockMatrix {
 public:
  typedef MatrixType SparseMatrixBlock;
  SparseBlockMatrix(const int* rbi, const int* cbi, int rb, int cb, bool hasStorage = true);
  SparseBlockMatrix();
  ~SparseBlockMatrix();
  void clear(bool dealloc = false);
  SparseMatrixBlock* block(int r, int c, bool alloc = false);
  const SparseMatrixBlock* block(int r, int c) const;
  int rowsOfBlock(int r) const;
  int colsOfBlock(int c) const;
  int rowBaseOfBlock(int r) const;
  int colBaseOfBlock(int c) const;
};
}  // namespace g2o
